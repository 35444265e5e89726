// Gate arithmetic of the convolutional GRU fuser (latentfusion/modules/gru.py:30-43, recon/fusion.py:188-197)
// for the inference path: the three gate convolutions read ONE channels-last record [x (Cx) | state (Ch)] per
// voxel, so the element-wise steps write straight into the state slots of that record instead of building
// new concatenations:
//   stage A:  u = sigmoid(ur[0:Ch]);  rec.state = h * sigmoid(ur[Ch:2Ch])          (input of the out gate)
//   stage B:  h' = h * (1 - u) + c * u;  rec.state = h'                               (input of the next step)
// (no tanh on the candidate: SURVEY Q13).  One thread per (voxel, channel); the arithmetic mirrors the ATen
// expression order (separate multiplies and add, no fma contraction).
#include "lf_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ void __launch_bounds__(256) gru_stage_a_kernel(const float* __restrict__ upre, const float* __restrict__ rpre,
                                                          int pre_stride, const float* __restrict__ h,
                                                          float* __restrict__ u, float* __restrict__ rec, long nvox, int Ch,
                                                          int rec_stride, int rec_off) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nvox * Ch) return;
  const long v = idx / Ch;
  const int c = (int)(idx - v * Ch);
  const float uu = sigmoidf_(upre[v * pre_stride + c]);
  const float rr = sigmoidf_(rpre[v * pre_stride + c]);
  u[idx] = uu;
  rec[v * rec_stride + rec_off + c] = __fmul_rn(h[idx], rr);
}

__global__ void __launch_bounds__(256) gru_stage_b_kernel(const float* __restrict__ h, const float* __restrict__ u,
                                                          const float* __restrict__ cand, float* __restrict__ h_out,
                                                          float* __restrict__ rec, long nvox, int Ch, int rec_stride,
                                                          int rec_off) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nvox * Ch) return;
  const long v = idx / Ch;
  const int c = (int)(idx - v * Ch);
  const float uu = u[idx];
  const float hn = __fadd_rn(__fmul_rn(h[idx], __fsub_rn(1.f, uu)), __fmul_rn(cand[idx], uu));
  h_out[idx] = hn;
  if (rec != nullptr) rec[v * rec_stride + rec_off + c] = hn;
}

// ---- backward of the two stages (training path; plain [n] arrays, 4 elements per thread) ----
//   stage B:  h' = h (1 - u) + c u          ->  gh = g (1 - u),  gu = g (c - h),  gc = g u
//   stage A:  u = s(upre), rh = h s(rpre)   ->  gupre = gu u (1 - u),  grpre = grh h r (1 - r),  gh = grh r
__global__ void __launch_bounds__(256) gru_stage_b_bwd_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ h,
                                                              const f32x4* __restrict__ u, const f32x4* __restrict__ cand,
                                                              f32x4* __restrict__ gh, f32x4* __restrict__ gu,
                                                              f32x4* __restrict__ gc, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 gg = g[i], uu = u[i];
  gh[i] = gg * (1.f - uu);
  gu[i] = gg * (cand[i] - h[i]);
  gc[i] = gg * uu;
}

__global__ void __launch_bounds__(256) gru_stage_a_bwd_kernel(const f32x4* __restrict__ gu, const f32x4* __restrict__ grh,
                                                              const f32x4* __restrict__ u, const f32x4* __restrict__ rpre,
                                                              const f32x4* __restrict__ h, f32x4* __restrict__ gupre,
                                                              f32x4* __restrict__ grpre, f32x4* __restrict__ gh, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 uu = u[i], rp = rpre[i], hh = h[i], gr = grh[i];
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = sigmoidf_(rp[e]);
  gupre[i] = gu[i] * uu * (1.f - uu);
  grpre[i] = gr * hh * r * (1.f - r);
  gh[i] = gr * r;
}

// ---- ConvLSTM cell arithmetic (latentfusion/modules/lstm.py:41-56): cc = conv([x, h]) holds the four gate
// pre-activations of a voxel as channel blocks [i | f | o | g] of Ch channels each (channels-last record of 4*Ch floats)
//   c' = sigmoid(f) c + sigmoid(i) tanh(g);   h' = sigmoid(o) tanh(c')
// one thread per (voxel, channel); backward recomputes the gates from cc.
__global__ void __launch_bounds__(256) lstm_cell_fwd_kernel(const float* __restrict__ cc, const float* __restrict__ c_cur,
                                                            float* __restrict__ h_next, float* __restrict__ c_next, long nvox, int Ch) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nvox * Ch) return;
  const long v = idx / Ch;
  const int c = (int)(idx - v * Ch);
  const float* r = cc + v * 4 * Ch + c;
  const float i = sigmoidf_(r[0]), f = sigmoidf_(r[Ch]), o = sigmoidf_(r[2 * Ch]), g = tanhf(r[3 * Ch]);
  const float cn = __fadd_rn(__fmul_rn(f, c_cur[idx]), __fmul_rn(i, g));
  c_next[idx] = cn;
  h_next[idx] = __fmul_rn(o, tanhf(cn));
}

// given gh = dL/dh', gcn = dL/dc' (from the next step; may be NULL): gcc (4*Ch per voxel) and gc = dL/dc
__global__ void __launch_bounds__(256) lstm_cell_bwd_kernel(const float* __restrict__ cc, const float* __restrict__ c_cur,
                                                            const float* __restrict__ gh, const float* __restrict__ gcn,
                                                            float* __restrict__ gcc, float* __restrict__ gc, long nvox, int Ch) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nvox * Ch) return;
  const long v = idx / Ch;
  const int c = (int)(idx - v * Ch);
  const float* r = cc + v * 4 * Ch + c;
  const float i = sigmoidf_(r[0]), f = sigmoidf_(r[Ch]), o = sigmoidf_(r[2 * Ch]), g = tanhf(r[3 * Ch]);
  const float cc_ = c_cur[idx];
  const float cn = f * cc_ + i * g;
  const float tc = tanhf(cn);
  const float ghv = gh ? gh[idx] : 0.f;
  const float dcn = (gcn ? gcn[idx] : 0.f) + ghv * o * (1.f - tc * tc);
  float* w = gcc + v * 4 * Ch + c;
  w[0] = dcn * g * i * (1.f - i);
  w[Ch] = dcn * cc_ * f * (1.f - f);
  w[2 * Ch] = ghv * tc * o * (1.f - o);
  w[3 * Ch] = dcn * i * (1.f - g * g);
  gc[idx] = dcn * f;
}


// ---- training path of the 16-channel ConvGRU fuser sequenced explicitly (round 5; ops._GruFuse): the recurrent state h, the
// incoming gradient chain and the per-gate gradient accumulators stay fp32; the gate pre-activations, h*r, the candidate and
// the gate gradients -- tensors that are produced and consumed once per step -- are stored as T = float or bf16 (the bf16
// autocast policy: they are outputs / operands of half-precision convolutions).  u and r are recomputed from the saved
// pre-activations, the two backward stages fold in what autograd ran as separate passes:
//   stage A    rh = h s(rpre)
//   stage B    h' = h (1 - u) + c u,                 u = s(upre)
//   stage B'   gh1 = g (1 - u),  gupre = g (c - h) u (1 - u),  gc = g u;          acc_u += gupre, acc_o += gc
//   stage A'   grpre = grh h r (1 - r),  gh12 = gh1 + grh r,  r = s(rpre);        acc_r += grpre
// (the accumulators hold sum over the views of the gate gradients: bias and coordinate-channel weight gradients need only it)
typedef __bf16 bf16x4g __attribute__((ext_vector_type(4)));
typedef unsigned u32x2g __attribute__((ext_vector_type(2)));
template <typename T> struct Rec4;
template <> struct Rec4<float> {
  static __device__ __forceinline__ f32x4 ld(const void* p, long i) { return ((const f32x4*)p)[i]; }
  static __device__ __forceinline__ void st(void* p, long i, f32x4 v) { ((f32x4*)p)[i] = v; }
};
template <> struct Rec4<__bf16> {
  static __device__ __forceinline__ f32x4 ld(const void* p, long i) { return __builtin_convertvector(((const bf16x4g*)p)[i], f32x4); }
  static __device__ __forceinline__ void st(void* p, long i, f32x4 v) { ((bf16x4g*)p)[i] = __builtin_convertvector(v, bf16x4g); }
};
__device__ __forceinline__ f32x4 sigmoid4(f32x4 v) {
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = sigmoidf_(v[e]);
  return r;
}

template <typename T>
__global__ void __launch_bounds__(256) gru_train_a_kernel(const void* __restrict__ rpre, const f32x4* __restrict__ h,
                                                          void* __restrict__ rh, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  Rec4<T>::st(rh, i, h[i] * sigmoid4(Rec4<T>::ld(rpre, i)));
}

template <typename T>
__global__ void __launch_bounds__(256) gru_train_b_kernel(const f32x4* __restrict__ h, const void* __restrict__ upre,
                                                          const void* __restrict__ cand, f32x4* __restrict__ h_out, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 u = sigmoid4(Rec4<T>::ld(upre, i)), c = Rec4<T>::ld(cand, i), hh = h[i];
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fmul_rn(hh[e], __fsub_rn(1.f, u[e])), __fmul_rn(c[e], u[e]));
  h_out[i] = o;
}

// (gupre may alias upre and gc may alias cand -- round 6 writes the gate gradients over the saved pre-activations: these four
// pointers carry no __restrict__, every load of a thread precedes its stores)
template <typename T>
__global__ void __launch_bounds__(256) gru_train_b_bwd_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ h,
                                                              const void* upre, const void* cand,
                                                              f32x4* __restrict__ gh1, void* gupre, void* gc,
                                                              f32x4* __restrict__ acc_u, f32x4* __restrict__ acc_o, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 gg = g[i], u = sigmoid4(Rec4<T>::ld(upre, i)), c = Rec4<T>::ld(cand, i), hh = h[i];
  const f32x4 gu = gg * (c - hh);
  const f32x4 gup = gu * u * (1.f - u), gcc = gg * u;
  gh1[i] = gg * (1.f - u);
  Rec4<T>::st(gupre, i, gup);
  Rec4<T>::st(gc, i, gcc);
  if (acc_u != nullptr) acc_u[i] += gup;
  if (acc_o != nullptr) acc_o[i] += gcc;
}

template <typename T>
__global__ void __launch_bounds__(256) gru_train_a_bwd_kernel(const void* __restrict__ grh, const void* __restrict__ rpre,
                                                              const f32x4* __restrict__ h, const f32x4* __restrict__ gh1,
                                                              void* __restrict__ grpre, f32x4* __restrict__ gh12,
                                                              f32x4* __restrict__ acc_r, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 gr = Rec4<T>::ld(grh, i), r = sigmoid4(Rec4<T>::ld(rpre, i)), hh = h[i];
  const f32x4 grp = gr * hh * r * (1.f - r);
  Rec4<T>::st(grpre, i, grp);
  gh12[i] = gh1[i] + gr * r;
  if (acc_r != nullptr) acc_r[i] += grp;
}

// dst[i] (+)= sum over the n volumes of src[v][i], in view order (deterministic): the sum over the recurrence's steps of a gate
// gradient, which is all the bias and the coordinate-channel weight gradients of that gate need
__global__ void __launch_bounds__(256) sum_views_bf16_kernel(const bf16x4g* __restrict__ src, f32x4* __restrict__ dst, long n4, int nv,
                                                             int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 s = accumulate ? dst[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
  int v = 0;
  for (; v + 4 <= nv; v += 4) {                                   // four loads in flight
    const bf16x4g a = src[(long)v * n4 + i], b = src[(long)(v + 1) * n4 + i], c = src[(long)(v + 2) * n4 + i], d = src[(long)(v + 3) * n4 + i];
    s += __builtin_convertvector(a, f32x4);
    s += __builtin_convertvector(b, f32x4);
    s += __builtin_convertvector(c, f32x4);
    s += __builtin_convertvector(d, f32x4);
  }
  for (; v < nv; ++v) s += __builtin_convertvector(src[(long)v * n4 + i], f32x4);
  dst[i] = s;
}

}  // namespace

extern "C" int lf_sum_views_bf16(const void* src, float* dst, long n, int views, int accumulate, void* stream) {
  lf_clear_error();
  if (n <= 0 || (n & 3) || views <= 0 || src == nullptr || dst == nullptr) return LF_EINVAL;
  if (!lf_aligned16(src) || !lf_aligned16(dst)) return LF_EALIGN;
  hipLaunchKernelGGL(sum_views_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16x4g*)src, (f32x4*)dst, n / 4, views, accumulate ? 1 : 0);
  return lf_launch_status();
}

extern "C" int lf_lstm_cell_fwd(const float* cc, const float* c_cur, float* h_next, float* c_next, long nvox, int Ch, void* stream) {
  lf_clear_error();
  if (nvox <= 0 || Ch <= 0 || nvox * Ch >= 0x7fffffff00L) return LF_EINVAL;
  hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3((unsigned)((nvox * Ch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cc, c_cur,
                     h_next, c_next, nvox, Ch);
  return lf_launch_status();
}

extern "C" int lf_lstm_cell_bwd(const float* cc, const float* c_cur, const float* gh, const float* gcn, float* gcc, float* gc,
                                long nvox, int Ch, void* stream) {
  lf_clear_error();
  if (nvox <= 0 || Ch <= 0 || nvox * Ch >= 0x7fffffff00L || (gh == nullptr && gcn == nullptr)) return LF_EINVAL;
  hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3((unsigned)((nvox * Ch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cc, c_cur,
                     gh, gcn, gcc, gc, nvox, Ch);
  return lf_launch_status();
}

extern "C" int lf_gru_stage_b_bwd(const float* g, const float* h, const float* u, const float* cand, float* gh, float* gu,
                                  float* gc, long n, void* stream) {
  lf_clear_error();
  if (n <= 0 || (n & 3)) return LF_EINVAL;
  if (!lf_aligned16(g) || !lf_aligned16(h) || !lf_aligned16(u) || !lf_aligned16(cand) || !lf_aligned16(gh) || !lf_aligned16(gu) ||
      !lf_aligned16(gc)) return LF_EALIGN;
  hipLaunchKernelGGL(gru_stage_b_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)g, (const f32x4*)h, (const f32x4*)u, (const f32x4*)cand, (f32x4*)gh, (f32x4*)gu, (f32x4*)gc, n / 4);
  return lf_launch_status();
}

extern "C" int lf_gru_stage_a_bwd(const float* gu, const float* grh, const float* u, const float* rpre, const float* h,
                                  float* gupre, float* grpre, float* gh, long n, void* stream) {
  lf_clear_error();
  if (n <= 0 || (n & 3)) return LF_EINVAL;
  if (!lf_aligned16(gu) || !lf_aligned16(grh) || !lf_aligned16(u) || !lf_aligned16(rpre) || !lf_aligned16(h) ||
      !lf_aligned16(gupre) || !lf_aligned16(grpre) || !lf_aligned16(gh)) return LF_EALIGN;
  hipLaunchKernelGGL(gru_stage_a_bwd_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)gu, (const f32x4*)grh, (const f32x4*)u, (const f32x4*)rpre, (const f32x4*)h, (f32x4*)gupre,
                     (f32x4*)grpre, (f32x4*)gh, n / 4);
  return lf_launch_status();
}

extern "C" int lf_gru_stage_a(const float* upre, const float* rpre, int pre_stride, const float* h, float* u, float* rec,
                              long nvox, int Ch, int rec_stride, int rec_off, void* stream) {
  lf_clear_error();
  if (nvox <= 0 || Ch <= 0 || rec_stride < rec_off + Ch || rec_off < 0 || nvox * Ch >= 0x7fffffff00L) return LF_EINVAL;
  if (pre_stride < Ch) return LF_EINVAL;
  hipLaunchKernelGGL(gru_stage_a_kernel, dim3((unsigned)((nvox * Ch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, upre, rpre,
                     pre_stride, h, u, rec, nvox, Ch, rec_stride, rec_off);
  return lf_launch_status();
}

extern "C" int lf_gru_stage_b(const float* h, const float* u, const float* cand, float* h_out, float* rec, long nvox, int Ch,
                              int rec_stride, int rec_off, void* stream) {
  lf_clear_error();
  if (nvox <= 0 || Ch <= 0 || nvox * Ch >= 0x7fffffff00L) return LF_EINVAL;
  if (rec != nullptr && (rec_stride < rec_off + Ch || rec_off < 0)) return LF_EINVAL;
  hipLaunchKernelGGL(gru_stage_b_kernel, dim3((unsigned)((nvox * Ch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h, u, cand,
                     h_out, rec, nvox, Ch, rec_stride, rec_off);
  return lf_launch_status();
}

// ---- training-path stages (see the kernels above); bf16 != 0: the T-typed tensors are bf16 ----
static inline bool gru_train_ok(long n, const void* const* ptrs, int np) {
  if (n <= 0 || (n & 3)) return false;
  for (int i = 0; i < np; ++i)
    if (ptrs[i] != nullptr && !lf_aligned16(ptrs[i])) return false;
  return true;
}
#define GRU_GRID(n) dim3((unsigned)(((n) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream

extern "C" int lf_gru_train_stage_a(const void* rpre, const float* h, void* rh, long n, int bf16, void* stream) {
  lf_clear_error();
  const void* ps[] = {rpre, h, rh};
  if (!gru_train_ok(n, ps, 3) || !rpre || !h || !rh) return LF_EINVAL;
  if (bf16) hipLaunchKernelGGL(gru_train_a_kernel<__bf16>, GRU_GRID(n), rpre, (const f32x4*)h, rh, n / 4);
  else hipLaunchKernelGGL(gru_train_a_kernel<float>, GRU_GRID(n), rpre, (const f32x4*)h, rh, n / 4);
  return lf_launch_status();
}

extern "C" int lf_gru_train_stage_b(const float* h, const void* upre, const void* cand, float* h_out, long n, int bf16, void* stream) {
  lf_clear_error();
  const void* ps[] = {h, upre, cand, h_out};
  if (!gru_train_ok(n, ps, 4) || !h || !upre || !cand || !h_out) return LF_EINVAL;
  if (bf16) hipLaunchKernelGGL(gru_train_b_kernel<__bf16>, GRU_GRID(n), (const f32x4*)h, upre, cand, (f32x4*)h_out, n / 4);
  else hipLaunchKernelGGL(gru_train_b_kernel<float>, GRU_GRID(n), (const f32x4*)h, upre, cand, (f32x4*)h_out, n / 4);
  return lf_launch_status();
}

extern "C" int lf_gru_train_stage_b_bwd(const float* g, const float* h, const void* upre, const void* cand, float* gh1, void* gupre,
                                        void* gc, float* acc_u, float* acc_o, long n, int bf16, void* stream) {
  lf_clear_error();
  const void* ps[] = {g, h, upre, cand, gh1, gupre, gc, acc_u, acc_o};
  if (!gru_train_ok(n, ps, 9) || !g || !h || !upre || !cand || !gh1 || !gupre || !gc) return LF_EINVAL;
  if (bf16) hipLaunchKernelGGL(gru_train_b_bwd_kernel<__bf16>, GRU_GRID(n), (const f32x4*)g, (const f32x4*)h, upre, cand, (f32x4*)gh1,
                               gupre, gc, (f32x4*)acc_u, (f32x4*)acc_o, n / 4);
  else hipLaunchKernelGGL(gru_train_b_bwd_kernel<float>, GRU_GRID(n), (const f32x4*)g, (const f32x4*)h, upre, cand, (f32x4*)gh1,
                          gupre, gc, (f32x4*)acc_u, (f32x4*)acc_o, n / 4);
  return lf_launch_status();
}

extern "C" int lf_gru_train_stage_a_bwd(const void* grh, const void* rpre, const float* h, const float* gh1, void* grpre, float* gh12,
                                        float* acc_r, long n, int bf16, void* stream) {
  lf_clear_error();
  const void* ps[] = {grh, rpre, h, gh1, grpre, gh12, acc_r};
  if (!gru_train_ok(n, ps, 7) || !grh || !rpre || !h || !gh1 || !grpre || !gh12) return LF_EINVAL;
  if (bf16) hipLaunchKernelGGL(gru_train_a_bwd_kernel<__bf16>, GRU_GRID(n), grh, rpre, (const f32x4*)h, (const f32x4*)gh1, grpre,
                               (f32x4*)gh12, (f32x4*)acc_r, n / 4);
  else hipLaunchKernelGGL(gru_train_a_bwd_kernel<float>, GRU_GRID(n), grh, rpre, (const f32x4*)h, (const f32x4*)gh1, grpre,
                          (f32x4*)gh12, (f32x4*)acc_r, n / 4);
  return lf_launch_status();
}
