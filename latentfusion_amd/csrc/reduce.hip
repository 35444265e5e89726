// Depth-column composites of the renderer and view reductions of the fusers (gfx950).
//
//   column sum            Photographer 'sum' projection            latentfusion/recon/models.py:436-437
//   column softmax        occlusion weights + expected depth       latentfusion/recon/models.py:378-395
//   column scale          z * depth_weights_resized                latentfusion/recon/models.py:427-430
//   view reductions       PoolFuser mean / max / abs_max / median  latentfusion/recon/fusion.py:45-57,
//                                                                  latentfusion/functional.py:47-49
//   view blend            BlendFuser softmax-over-views + sum      latentfusion/recon/fusion.py:139-148
//
// All of them are HBM-bound streaming passes over channels-last volumes [N][D][H][W][C]: a depth column
// (fixed h, w) is NOT contiguous -- consecutive depth samples are H*W*C floats apart -- so lanes run along the
// contiguous (w, c) axis (every wave-level load is one 1 KiB row segment) and the depth axis is split over the
// four waves of a workgroup (and, for the single-channel softmax, over four 16-lane groups of each wave, whose
// partial maxima / sums meet through wave shuffles); the cross-wave combine goes through LDS in a fixed order,
// so every result is deterministic.  No float atomics.
#include "lf_common.h"

namespace {

// i-th point of torch.linspace(-1, 1, n): start + step*i in the first half, end - step*(n-1-i) in the second
// (ATen's symmetric formula; step = 2/(n-1) rounded to fp32 on the host)
__device__ __forceinline__ float depth_coord(int i, int n, float step) {
  if (n == 1) return -1.f;                                       // linspace(-1, 1, 1) = [start]
  return (i < n / 2) ? -1.f + step * (float)i : 1.f - step * (float)(n - 1 - i);
}

// ---------------------------------------------------------------------------------------------------------
// column sum: x [N][D][R4] float4 -> y [N][R4];  workgroup = 64 columns x 4 depth phases (one per wave)
__global__ void __launch_bounds__(256) column_sum_fwd_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y,
                                                             int D, long R4) {
  __shared__ f32x4 part[3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 64 + lane;
  const long n = blockIdx.y;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (col < R4) {
    const f32x4* p = x + (n * D) * R4 + col;
    int d = w;
    // four independent loads in flight per lane
    for (; d + 12 < D; d += 16) {
      const f32x4 a = __builtin_nontemporal_load(p + (long)d * R4), b = __builtin_nontemporal_load(p + (long)(d + 4) * R4);
      const f32x4 c = __builtin_nontemporal_load(p + (long)(d + 8) * R4), e = __builtin_nontemporal_load(p + (long)(d + 12) * R4);
      acc += a; acc += b; acc += c; acc += e;
    }
    for (; d < D; d += 4) acc += __builtin_nontemporal_load(p + (long)d * R4);
  }
  if (w > 0) part[w - 1][lane] = acc;
  __syncthreads();
  if (w == 0 && col < R4) y[n * R4 + col] = ((acc + part[0][lane]) + part[1][lane]) + part[2][lane];
}

__global__ void __launch_bounds__(256) column_sum_fwd_scalar(const float* __restrict__ x, float* __restrict__ y, int D, long R) {
  const long col = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = blockIdx.y;
  if (col >= R) return;
  float acc = 0.f;
  for (int d = 0; d < D; ++d) acc += x[(n * D + d) * R + col];
  y[n * R + col] = acc;
}

// adjoint: gx[n][d][r] = gy[n][r]
__global__ void __launch_bounds__(256) column_sum_bwd_kernel(const f32x4* __restrict__ gy, f32x4* __restrict__ gx, int D, long R4) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const long col = (long)blockIdx.x * 64 + lane;
  const long n = blockIdx.y;
  if (col >= R4) return;
  const f32x4 g = gy[n * R4 + col];
  f32x4* p = gx + (n * D) * R4 + col;
  for (int d = w; d < D; d += 4) __builtin_nontemporal_store(g, p + (long)d * R4);
}

__global__ void __launch_bounds__(256) column_sum_bwd_scalar(const float* __restrict__ gy, float* __restrict__ gx, int D, long R) {
  const long col = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = blockIdx.y;
  if (col >= R) return;
  const float g = gy[n * R + col];
  for (int d = 0; d < D; ++d) gx[(n * D + d) * R + col] = g;
}

// ---------------------------------------------------------------------------------------------------------
// column softmax of single-channel logits [N][D][P] (+ expected depth).  Workgroup = 16 columns x 16 depth
// slices: lane = slice_in_wave * 16 + column, wave w owns slices 4w .. 4w+3.  A reduction over the column =
// two wave shuffles (xor 16, 32) + a 4-entry LDS exchange, always combined in the same order.
struct ColRed {
  float (*lds)[16];                                              // [4 waves][16 columns]
  int lane, w, col;
};

template <bool MAX>
__device__ __forceinline__ float column_allreduce(float v, const ColRed& cr) {
  // across the four 16-lane groups of the wave
  const float o1 = __shfl_xor(v, 16, 64);
  v = MAX ? fmaxf(v, o1) : v + o1;
  const float o2 = __shfl_xor(v, 32, 64);
  v = MAX ? fmaxf(v, o2) : v + o2;
  __syncthreads();                                               // previous use of the LDS slots is over
  if (cr.lane < 16) cr.lds[cr.w][cr.col] = v;
  __syncthreads();
  const float a = cr.lds[0][cr.col], b = cr.lds[1][cr.col], c = cr.lds[2][cr.col], d = cr.lds[3][cr.col];
  return MAX ? fmaxf(fmaxf(a, b), fmaxf(c, d)) : (a + b) + (c + d);
}

__global__ void __launch_bounds__(256) column_softmax_fwd_kernel(const float* __restrict__ logits, float* __restrict__ wout,
                                                                 float* __restrict__ zdepth, int D, long P, float step) {
  __shared__ float lds[4][16];
  ColRed cr;
  cr.lds = lds; cr.lane = threadIdx.x & 63; cr.w = threadIdx.x >> 6; cr.col = cr.lane & 15;
  const int slice = cr.w * 4 + (cr.lane >> 4);
  const long p = (long)blockIdx.x * 16 + cr.col;
  const long n = blockIdx.y;
  const bool live = p < P;
  const float* src = logits + (n * D) * P + (live ? p : 0);
  // pass 1: column maximum; pass 2: sum of exp(x - max); pass 3: normalised weights and their first moment
  float m = -INFINITY;
  if (live) for (int d = slice; d < D; d += 16) m = fmaxf(m, src[(long)d * P]);
  m = column_allreduce<true>(m, cr);
  float s = 0.f;
  if (live) for (int d = slice; d < D; d += 16) s += expf(src[(long)d * P] - m);
  s = column_allreduce<false>(s, cr);
  float mom = 0.f;
  if (live) {
    float* dst = wout ? wout + (n * D) * P + p : nullptr;
    for (int d = slice; d < D; d += 16) {
      const float wv = expf(src[(long)d * P] - m) / s;
      if (dst) dst[(long)d * P] = wv;
      mom += depth_coord(d, D, step) * wv;
    }
  }
  if (zdepth != nullptr) {
    mom = column_allreduce<false>(mom, cr);
    if (live && threadIdx.x < 16) zdepth[n * P + p] = mom;
  }
}

// The same with the logits formed on the way in (round 6): logit[n][d][p] = (sum_c y[n][d][p][c] * w[c]) * he + bias, the occlusion
// module's 16 -> 1 output block (reference modules/blocks.py OutputBlock without activation) over a channels-last 16-channel
// volume.  Every record is read ONCE (a thread keeps the <= 16 logits of its depth slices in registers: D <= 256) and the
// logits volume is never written -- the pointwise launch that produced it read the same 64 bytes per voxel for 4 bytes out.
constexpr int CSH_MAX = 16;                                        // depth values per thread at most
__global__ void __launch_bounds__(256) column_softmax_head_fwd_kernel(const f32x4* __restrict__ y, const float* __restrict__ w16,
                                                                      const float* __restrict__ bias, float he,
                                                                      float* __restrict__ wout, float* __restrict__ zdepth,
                                                                      int D, long P, float step) {
  __shared__ float lds[4][16];
  ColRed cr;
  cr.lds = lds; cr.lane = threadIdx.x & 63; cr.w = threadIdx.x >> 6; cr.col = cr.lane & 15;
  const int slice = cr.w * 4 + (cr.lane >> 4);
  const long p = (long)blockIdx.x * 16 + cr.col;
  const long n = blockIdx.y;
  const bool live = p < P;
  const f32x4 w0 = *(const f32x4*)w16, w1 = *(const f32x4*)(w16 + 4), w2 = *(const f32x4*)(w16 + 8), w3 = *(const f32x4*)(w16 + 12);
  const float b = bias != nullptr ? bias[0] : 0.f;
  const f32x4* src = y + ((n * D) * P + (live ? p : 0)) * 4;
  float lg[CSH_MAX];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < CSH_MAX; ++i) {
    const int d = slice + 16 * i;
    lg[i] = -INFINITY;
    if (live && d < D) {
      const f32x4* r = src + (long)d * P * 4;
      const f32x4 a0 = r[0], a1 = r[1], a2 = r[2], a3 = r[3];
      float sum = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += a0[c] * w0[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += a1[c] * w1[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += a2[c] * w2[c];
#pragma unroll
      for (int c = 0; c < 4; ++c) sum += a3[c] * w3[c];
      lg[i] = sum * he + b;
      m = fmaxf(m, lg[i]);
    }
  }
  m = column_allreduce<true>(m, cr);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CSH_MAX; ++i)
    if (live && slice + 16 * i < D) { lg[i] = expf(lg[i] - m); s += lg[i]; }
  s = column_allreduce<false>(s, cr);
  float mom = 0.f;
  if (live) {
    float* dst = wout ? wout + (n * D) * P + p : nullptr;
#pragma unroll
    for (int i = 0; i < CSH_MAX; ++i) {
      const int d = slice + 16 * i;
      if (d < D) {
        const float wv = lg[i] / s;
        if (dst) dst[(long)d * P] = wv;
        mom += depth_coord(d, D, step) * wv;
      }
    }
  }
  if (zdepth != nullptr) {
    mom = column_allreduce<false>(mom, cr);
    if (live && threadIdx.x < 16) zdepth[n * P + p] = mom;
  }
}

// glogits_d = w_d (t_d - sum_e w_e t_e),  t_d = gw_d + gz * coord_d
__global__ void __launch_bounds__(256) column_softmax_bwd_kernel(const float* __restrict__ wts, const float* __restrict__ gw,
                                                                 const float* __restrict__ gz, float* __restrict__ glogits,
                                                                 int D, long P, float step) {
  __shared__ float lds[4][16];
  ColRed cr;
  cr.lds = lds; cr.lane = threadIdx.x & 63; cr.w = threadIdx.x >> 6; cr.col = cr.lane & 15;
  const int slice = cr.w * 4 + (cr.lane >> 4);
  const long p = (long)blockIdx.x * 16 + cr.col;
  const long n = blockIdx.y;
  const bool live = p < P;
  const long base = (n * D) * P + (live ? p : 0);
  const float g1 = (live && gz != nullptr) ? gz[n * P + p] : 0.f;
  float dot = 0.f;
  if (live)
    for (int d = slice; d < D; d += 16) {
      const float t = (gw ? gw[base + (long)d * P] : 0.f) + g1 * depth_coord(d, D, step);
      dot += wts[base + (long)d * P] * t;
    }
  dot = column_allreduce<false>(dot, cr);
  if (live)
    for (int d = slice; d < D; d += 16) {
      const float t = (gw ? gw[base + (long)d * P] : 0.f) + g1 * depth_coord(d, D, step);
      glogits[base + (long)d * P] = wts[base + (long)d * P] * (t - dot);
    }
}

// ---------------------------------------------------------------------------------------------------------
// out[row][:] = z[row][:] * w[row]   (rows = N*D*H*W voxels, C channels, channels-last)
__global__ void __launch_bounds__(256) column_scale_fwd_kernel(const f32x4* __restrict__ z, const float* __restrict__ w,
                                                               f32x4* __restrict__ out, long n4, int c4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  out[i] = z[i] * w[i / c4];
}

// gz = gout * w;  gw[row] = sum_c gout * z.  `lpr` lanes (a power of two <= 64) cooperate on one row.
__global__ void __launch_bounds__(256) column_scale_bwd_kernel(const f32x4* __restrict__ gout, const f32x4* __restrict__ z,
                                                               const float* __restrict__ w, f32x4* __restrict__ gz,
                                                               float* __restrict__ gw, long rows, int c4, int lpr) {
  const int rpb = 256 / lpr;
  const long row = (long)blockIdx.x * rpb + threadIdx.x / lpr;
  const int q = threadIdx.x % lpr;
  const bool live = row < rows;                                  // (whole lpr-lane groups are live or dead together)
  float dot = 0.f;
  if (live) {
    const float wv = w[row];
    for (int j = q; j < c4; j += lpr) {
      const f32x4 g = gout[row * c4 + j];
      if (gw != nullptr) {
        const f32x4 zz = z[row * c4 + j];
        dot += (g[0] * zz[0] + g[1] * zz[1]) + (g[2] * zz[2] + g[3] * zz[3]);
      }
      if (gz != nullptr) gz[row * c4 + j] = g * wv;
    }
  }
  for (int o = lpr >> 1; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 64);
  if (live && gw != nullptr && q == 0) gw[row] = dot;
}

// ---------------------------------------------------------------------------------------------------------
// reductions over the view axis: z [V][n] (views `vstride` floats apart) -> out [n] (+ selected view index)
template <typename T> struct VecOps;
template <> struct VecOps<float> { enum { W = 1 }; static __device__ float get(float v, int) { return v; }
                                   static __device__ void set(float& v, int, float x) { v = x; } };
template <> struct VecOps<f32x4> { enum { W = 4 }; static __device__ float get(const f32x4& v, int e) { return v[e]; }
                                   static __device__ void set(f32x4& v, int e, float x) { v[e] = x; } };

template <typename T>
__global__ void __launch_bounds__(256) fuse_views_fwd_kernel(const T* __restrict__ z, T* __restrict__ out, int* __restrict__ idx,
                                                             int kind, int V, long nvec, long vstride_vec) {
  typedef VecOps<T> O;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nvec) return;
  T r;
  int sel[O::W];
  // views are read four at a time (independent streamed loads: every view is a separate stream `vstride` apart, one
  // load in flight per lane would leave the kernel latency-bound) and combined in view order
  if (kind == LF_FUSE_MEAN) {
    T acc = __builtin_nontemporal_load(z + i);
    int v = 1;
    for (; v + 3 < V; v += 4) {
      const T c0 = __builtin_nontemporal_load(z + i + (long)v * vstride_vec), c1 = __builtin_nontemporal_load(z + i + (long)(v + 1) * vstride_vec);
      const T c2 = __builtin_nontemporal_load(z + i + (long)(v + 2) * vstride_vec), c3 = __builtin_nontemporal_load(z + i + (long)(v + 3) * vstride_vec);
      acc += c0; acc += c1; acc += c2; acc += c3;
    }
    for (; v < V; ++v) acc += __builtin_nontemporal_load(z + i + (long)v * vstride_vec);
    r = acc / (float)V;
  } else if (kind == LF_FUSE_MAX || kind == LF_FUSE_ABSMAX) {
    r = __builtin_nontemporal_load(z + i);
#pragma unroll
    for (int e = 0; e < O::W; ++e) sel[e] = 0;
    auto consider = [&](const T& c, int v) {
#pragma unroll
      for (int e = 0; e < O::W; ++e) {
        const float a = O::get(c, e), b = O::get(r, e);
        // first maximum wins ties; a NaN in ANY view wins over every number and is then kept (torch.max / abs().max()
        // propagate NaN, and so do the host and the sharded all-reduce forms of the same pool)
        const bool take = ((kind == LF_FUSE_MAX) ? (a > b) : (fabsf(a) > fabsf(b))) || (a != a && b == b);
        if (take) { O::set(r, e, a); sel[e] = v; }
      }
    };
    int v = 1;
    for (; v + 3 < V; v += 4) {
      const T c0 = __builtin_nontemporal_load(z + i + (long)v * vstride_vec), c1 = __builtin_nontemporal_load(z + i + (long)(v + 1) * vstride_vec);
      const T c2 = __builtin_nontemporal_load(z + i + (long)(v + 2) * vstride_vec), c3 = __builtin_nontemporal_load(z + i + (long)(v + 3) * vstride_vec);
      consider(c0, v); consider(c1, v + 1); consider(c2, v + 2); consider(c3, v + 3);
    }
    for (; v < V; ++v) consider(__builtin_nontemporal_load(z + i + (long)v * vstride_vec), v);
  } else {
    // lower median = element of rank (V-1)/2 (torch.median); rank by counting, ties by view index
    const int want = (V - 1) / 2;
#pragma unroll
    for (int e = 0; e < O::W; ++e) sel[e] = 0;
    r = z[i];
    for (int v = 0; v < V; ++v) {
      const T c = z[i + v * vstride_vec];
      int rank[O::W];
#pragma unroll
      for (int e = 0; e < O::W; ++e) rank[e] = 0;
      for (int u = 0; u < V; ++u) {
        const T o = z[i + u * vstride_vec];                        // L1-resident re-read
#pragma unroll
        for (int e = 0; e < O::W; ++e) {
          const float a = O::get(o, e), b = O::get(c, e);
          rank[e] += (a < b || (a == b && u < v)) ? 1 : 0;
        }
      }
#pragma unroll
      for (int e = 0; e < O::W; ++e)
        if (rank[e] == want) { O::set(r, e, O::get(c, e)); sel[e] = v; }
    }
  }
  out[i] = r;
  if (idx != nullptr && kind != LF_FUSE_MEAN) {
#pragma unroll
    for (int e = 0; e < O::W; ++e) idx[i * O::W + e] = sel[e];
  }
}

// adjoint: mean -> g / V to every view; selections -> g to the selected view, 0 elsewhere
__global__ void __launch_bounds__(256) fuse_views_bwd_kernel(const float* __restrict__ g, const int* __restrict__ idx,
                                                             float* __restrict__ gz, int kind, int V, long n, long vstride) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float gg = g[i];
  if (kind == LF_FUSE_MEAN) {
    const float s = gg / (float)V;
    for (int v = 0; v < V; ++v) gz[i + v * vstride] = s;
  } else {
    const int s = idx[i];
    for (int v = 0; v < V; ++v) gz[i + v * vstride] = (v == s) ? gg : 0.f;
  }
}

// blend: w = softmax_v(logits[v][row]);  out[row][:] = sum_v z[v][row][:] * w[v][row]
__global__ void __launch_bounds__(256) fuse_blend_fwd_kernel(const f32x4* __restrict__ z, const float* __restrict__ logits,
                                                             float* __restrict__ w, f32x4* __restrict__ out, int V,
                                                             long rows, int c4, long zstride_vec, long lstride) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * c4) return;
  const long row = i / c4;
  const int q = (int)(i - row * c4);
  float m = -INFINITY;
  for (int v = 0; v < V; ++v) m = fmaxf(m, logits[row + v * lstride]);
  float s = 0.f;
  for (int v = 0; v < V; ++v) s += expf(logits[row + v * lstride] - m);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int v = 0; v < V; ++v) {
    const float wv = expf(logits[row + v * lstride] - m) / s;
    if (q == 0 && w != nullptr) w[row + v * lstride] = wv;
    acc += z[i + v * zstride_vec] * wv;
  }
  out[i] = acc;
}

// gz[v] = g * w_v;  glogits_v = w_v (a_v - sum_u w_u a_u),  a_v = sum_c g * z_v
__global__ void __launch_bounds__(256) fuse_blend_bwd_kernel(const f32x4* __restrict__ g, const f32x4* __restrict__ z,
                                                             const float* __restrict__ w, f32x4* __restrict__ gz,
                                                             float* __restrict__ glogits, int V, long rows, int c4, int lpr,
                                                             long zstride_vec, long lstride) {
  const int rpb = 256 / lpr;
  const long row = (long)blockIdx.x * rpb + threadIdx.x / lpr;
  const int q = threadIdx.x % lpr;
  const bool live = row < rows;
  float sdot = 0.f;
  for (int v = 0; v < V; ++v) {
    float a = 0.f;
    const float wv = live ? w[row + v * lstride] : 0.f;
    if (live)
      for (int j = q; j < c4; j += lpr) {
        const f32x4 gg = g[row * c4 + j];
        const f32x4 zz = z[row * c4 + j + v * zstride_vec];
        a += (gg[0] * zz[0] + gg[1] * zz[1]) + (gg[2] * zz[2] + gg[3] * zz[3]);
        if (gz != nullptr) gz[row * c4 + j + v * zstride_vec] = gg * wv;
      }
    for (int o = lpr >> 1; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    sdot += wv * a;
    if (live && q == 0 && glogits != nullptr) glogits[row + v * lstride] = a;      // parked; fixed up below by the same lane
  }
  if (live && q == 0 && glogits != nullptr)
    for (int v = 0; v < V; ++v) {
      const float a = glogits[row + v * lstride];
      glogits[row + v * lstride] = w[row + v * lstride] * (a - sdot);
    }
}

int lanes_per_row(int c4) {
  int l = 1;
  while (l < c4 && l < 64) l <<= 1;
  return l;
}

float lin_step(int D) { return D > 1 ? 2.0f / (float)(D - 1) : 0.f; }

}  // namespace

extern "C" int lf_column_reduce_sum_fwd(const float* x, float* y, int N, int D, long P, int C, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || C <= 0 || N > 65535) return LF_EINVAL;
  const long R = P * C;
  hipStream_t s = (hipStream_t)stream;
  if (R % 4 == 0 && lf_aligned16(x) && lf_aligned16(y)) {
    const long R4 = R / 4;
    hipLaunchKernelGGL(column_sum_fwd_kernel, dim3((unsigned)((R4 + 63) / 64), N), dim3(256), 0, s, (const f32x4*)x, (f32x4*)y, D, R4);
  } else {
    hipLaunchKernelGGL(column_sum_fwd_scalar, dim3((unsigned)((R + 255) / 256), N), dim3(256), 0, s, x, y, D, R);
  }
  return lf_launch_status();
}

extern "C" int lf_column_reduce_sum_bwd(const float* gy, float* gx, int N, int D, long P, int C, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || C <= 0 || N > 65535) return LF_EINVAL;
  const long R = P * C;
  hipStream_t s = (hipStream_t)stream;
  if (R % 4 == 0 && lf_aligned16(gy) && lf_aligned16(gx)) {
    const long R4 = R / 4;
    hipLaunchKernelGGL(column_sum_bwd_kernel, dim3((unsigned)((R4 + 63) / 64), N), dim3(256), 0, s, (const f32x4*)gy, (f32x4*)gx, D, R4);
  } else {
    hipLaunchKernelGGL(column_sum_bwd_scalar, dim3((unsigned)((R + 255) / 256), N), dim3(256), 0, s, gy, gx, D, R);
  }
  return lf_launch_status();
}

extern "C" int lf_column_softmax_fwd(const float* logits, float* weights, float* zdepth, int N, int D, long P, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || N > 65535 || (weights == nullptr && zdepth == nullptr)) return LF_EINVAL;
  hipLaunchKernelGGL(column_softmax_fwd_kernel, dim3((unsigned)((P + 15) / 16), N), dim3(256), 0, (hipStream_t)stream, logits,
                     weights, zdepth, D, P, lin_step(D));
  return lf_launch_status();
}

extern "C" int lf_column_softmax_head_fwd(const float* y, const float* w16, const float* bias, float he, float* weights, float* zdepth,
                                         int N, int D, long P, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || N > 65535 || D > 16 * CSH_MAX || y == nullptr || w16 == nullptr ||
      (weights == nullptr && zdepth == nullptr)) return LF_EINVAL;
  if (!lf_aligned16(y) || !lf_aligned16(w16)) return LF_EALIGN;
  hipLaunchKernelGGL(column_softmax_head_fwd_kernel, dim3((unsigned)((P + 15) / 16), N), dim3(256), 0, (hipStream_t)stream, (const f32x4*)y,
                     w16, bias, he, weights, zdepth, D, P, lin_step(D));
  return lf_launch_status();
}

extern "C" int lf_column_softmax_bwd(const float* weights, const float* gweights, const float* gzdepth, float* glogits,
                                     int N, int D, long P, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || N > 65535 || (gweights == nullptr && gzdepth == nullptr)) return LF_EINVAL;
  hipLaunchKernelGGL(column_softmax_bwd_kernel, dim3((unsigned)((P + 15) / 16), N), dim3(256), 0, (hipStream_t)stream, weights,
                     gweights, gzdepth, glogits, D, P, lin_step(D));
  return lf_launch_status();
}

extern "C" int lf_column_scale_fwd(const float* z, const float* w, float* out, long rows, int C, void* stream) {
  lf_clear_error();
  if (rows <= 0 || C <= 0 || (C & 3)) return LF_EINVAL;
  if (!lf_aligned16(z) || !lf_aligned16(out)) return LF_EALIGN;
  const long n4 = rows * (C / 4);
  hipLaunchKernelGGL(column_scale_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)z, w, (f32x4*)out, n4, C / 4);
  return lf_launch_status();
}

extern "C" int lf_column_scale_bwd(const float* gout, const float* z, const float* w, float* gz, float* gw, long rows, int C,
                                   void* stream) {
  lf_clear_error();
  if (rows <= 0 || C <= 0 || (C & 3) || (gz == nullptr && gw == nullptr)) return LF_EINVAL;
  if (!lf_aligned16(gout) || !lf_aligned16(z) || (gz && !lf_aligned16(gz))) return LF_EALIGN;
  const int c4 = C / 4, lpr = lanes_per_row(c4), rpb = 256 / lpr;
  hipLaunchKernelGGL(column_scale_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)gout, (const f32x4*)z, w, (f32x4*)gz, gw, rows, c4, lpr);
  return lf_launch_status();
}

extern "C" int lf_fuse_views_fwd(const float* z, float* out, int* idx, int kind, int V, long n, long view_stride, void* stream) {
  lf_clear_error();
  if (V <= 0 || n <= 0 || kind < LF_FUSE_MEAN || kind > LF_FUSE_MEDIAN || (V > 1 && view_stride < n)) return LF_EINVAL;
  if (kind == LF_FUSE_MEDIAN && V > 1024) return LF_EINVAL;          // rank counting is O(V^2) per element: any V works, slowly
  hipStream_t s = (hipStream_t)stream;
  if (n % 4 == 0 && view_stride % 4 == 0 && lf_aligned16(z) && lf_aligned16(out) && (idx == nullptr || lf_aligned16(idx))) {
    const long nv = n / 4;
    hipLaunchKernelGGL((fuse_views_fwd_kernel<f32x4>), dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s, (const f32x4*)z,
                       (f32x4*)out, idx, kind, V, nv, view_stride / 4);
  } else {
    hipLaunchKernelGGL((fuse_views_fwd_kernel<float>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, z, out, idx, kind, V, n,
                       view_stride);
  }
  return lf_launch_status();
}

extern "C" int lf_fuse_views_bwd(const float* g, const int* idx, float* gz, int kind, int V, long n, long view_stride, void* stream) {
  lf_clear_error();
  if (V <= 0 || n <= 0 || kind < LF_FUSE_MEAN || kind > LF_FUSE_MEDIAN || (V > 1 && view_stride < n)) return LF_EINVAL;
  if (kind != LF_FUSE_MEAN && idx == nullptr) return LF_EINVAL;
  hipLaunchKernelGGL(fuse_views_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, idx, gz, kind,
                     V, n, view_stride);
  return lf_launch_status();
}

extern "C" int lf_fuse_blend_fwd(const float* z, const float* logits, float* weights, float* out, int V, long rows, int C,
                                 long z_view_stride, long logit_view_stride, void* stream) {
  lf_clear_error();
  if (V <= 0 || rows <= 0 || C <= 0 || (C & 3) || (z_view_stride & 3)) return LF_EINVAL;
  if (!lf_aligned16(z) || !lf_aligned16(out)) return LF_EALIGN;
  const long n4 = rows * (C / 4);
  hipLaunchKernelGGL(fuse_blend_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)z,
                     logits, weights, (f32x4*)out, V, rows, C / 4, z_view_stride / 4, logit_view_stride);
  return lf_launch_status();
}

extern "C" int lf_fuse_blend_bwd(const float* g, const float* z, const float* weights, float* gz, float* glogits, int V, long rows,
                                 int C, long z_view_stride, long logit_view_stride, void* stream) {
  lf_clear_error();
  if (V <= 0 || rows <= 0 || C <= 0 || (C & 3) || (z_view_stride & 3) || (gz == nullptr && glogits == nullptr)) return LF_EINVAL;
  if (!lf_aligned16(g) || !lf_aligned16(z) || (gz && !lf_aligned16(gz))) return LF_EALIGN;
  const int c4 = C / 4, lpr = lanes_per_row(c4), rpb = 256 / lpr;
  hipLaunchKernelGGL(fuse_blend_bwd_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)g, (const f32x4*)z, weights, (f32x4*)gz, glogits, V, rows, c4, lpr, z_view_stride / 4,
                     logit_view_stride);
  return lf_launch_status();
}
