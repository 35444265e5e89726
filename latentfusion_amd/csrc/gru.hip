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

}  // namespace

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
