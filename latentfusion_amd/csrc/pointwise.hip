// HBM-bound row kernels of the reconstruct-and-render path (gfx950):
//   PixelNorm forward          latentfusion/modules/__init__.py:14-15
//   fused-epilogue backward    LeakyReLU' * PixelNorm' (autograd of modules/blocks.py:152-158)
//   layout changes             NCHW <-> NHWC, FactorProjection2d3d's view (modules/geometry.py:728)
// Rows are channels-last records of C floats.  When C/4 is a power of two <= 64 a group of C/4
// lanes owns one row (float4 per lane, xor-shuffle reduction inside the group); otherwise one
// wavefront walks the row.
#include "lf_common.h"

namespace {

__device__ __forceinline__ float group_sum(float v, int lanes) {
  for (int o = lanes >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// MODE 0: pixelnorm fwd (a = x).  MODE 1: epilogue bwd (a = gy, b = y, nrm in).
template <int MODE>
__global__ void __launch_bounds__(256) rows_vec4_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ nrm_in,
    float* __restrict__ out, float* __restrict__ nrm_out, long rows, int C, int lpr,
    unsigned flags, float slope, float eps) {
  const int rpb = 256 / lpr;                              // rows per block-iteration
  const int q = threadIdx.x % lpr, slot = threadIdx.x / lpr;
  const long iters = (rows + (long)gridDim.x * rpb - 1) / ((long)gridDim.x * rpb);
  for (long it = 0; it < iters; ++it) {
    const long row = (it * gridDim.x + blockIdx.x) * rpb + slot;
    const bool live = row < rows;
    f32x4 va = (f32x4){0.f, 0.f, 0.f, 0.f}, vb = va;
    if (live) {
      va = *(const f32x4*)(a + row * C + q * 4);
      if (MODE == 1) vb = *(const f32x4*)(b + row * C + q * 4);
    }
    if (MODE == 0) {
      float ss = va[0] * va[0] + va[1] * va[1] + va[2] * va[2] + va[3] * va[3];
      ss = group_sum(ss, lpr);
      const float r = sqrtf(ss / (float)C + eps);
      if (live) {
        f32x4 o = {va[0] / r, va[1] / r, va[2] / r, va[3] / r};
        *(f32x4*)(out + row * C + q * 4) = o;
        if (q == 0 && nrm_out) nrm_out[row] = r;
      }
    } else {
      f32x4 g = va;
      if (flags & LF_EPI_PIXELNORM) {
        float dot = va[0] * vb[0] + va[1] * vb[1] + va[2] * vb[2] + va[3] * vb[3];
        dot = group_sum(dot, lpr) / (float)C;
        const float r = live ? nrm_in[row] : 1.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = (va[e] - vb[e] * dot) / r;
      }
      if (flags & LF_EPI_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = vb[e] > 0.f ? g[e] : g[e] * slope;
      }
      if (live) *(f32x4*)(out + row * C + q * 4) = g;
    }
  }
}

// generic C: one wavefront per row
template <int MODE>
__global__ void __launch_bounds__(256) rows_wave_kernel(
    const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ nrm_in,
    float* __restrict__ out, float* __restrict__ nrm_out, long rows, int C,
    unsigned flags, float slope, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* pa = a + row * C;
    const float* pb = (MODE == 1) ? b + row * C : nullptr;
    float s = 0.f;
    if (MODE == 0) {
      for (int c = lane; c < C; c += 64) s += pa[c] * pa[c];
      s = lf_wave_sum(s);
      const float r = sqrtf(s / (float)C + eps);
      for (int c = lane; c < C; c += 64) out[row * C + c] = pa[c] / r;
      if (lane == 0 && nrm_out) nrm_out[row] = r;
    } else {
      float dot = 0.f, r = 1.f;
      if (flags & LF_EPI_PIXELNORM) {
        for (int c = lane; c < C; c += 64) s += pa[c] * pb[c];
        dot = lf_wave_sum(s) / (float)C;
        r = nrm_in[row];
      }
      for (int c = lane; c < C; c += 64) {
        float g = pa[c];
        const float yv = pb[c];
        if (flags & LF_EPI_PIXELNORM) g = (g - yv * dot) / r;
        if (flags & LF_EPI_LRELU) g = yv > 0.f ? g : g * slope;
        out[row * C + c] = g;
      }
    }
  }
}

// [N][C][P] -> [N][P][C] through a 32x32 LDS tile (coalesced on both sides)
__global__ void __launch_bounds__(256) transpose_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                        int R, long Q) {
  // src: [n][R][Q] -> dst: [n][Q][R]
  __shared__ float t[32][33];
  const int n = blockIdx.z;
  const long q0 = (long)blockIdx.x * 32;
  const int r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 32 x 8
  const float* s = src + (long)n * R * Q;
  float* d = dst + (long)n * R * Q;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i;
    const long q = q0 + tx;
    t[i][tx] = (r < R && q < Q) ? s[(long)r * Q + q] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const long q = q0 + i;
    const int r = r0 + tx;
    if (r < R && q < Q) d[q * R + r] = t[tx][i];
  }
}

// src [N][P][C0*S] (channel = c*S + d)  ->  dst [N][S][P][C0], optional per-row 1/norm scaling
__global__ void __launch_bounds__(256) lift_unfold_kernel(const float* __restrict__ src,
                                                          const float* __restrict__ nrm,
                                                          float* __restrict__ dst, long P, int C0, int S) {
  extern __shared__ float row[];                            // C0*S floats of one pixel
  const int n = blockIdx.y;
  const int CS = C0 * S;
  for (long p = blockIdx.x; p < P; p += gridDim.x) {
    const float* s = src + ((long)n * P + p) * CS;
    __syncthreads();
    for (int i = threadIdx.x; i < CS; i += 256) row[i] = nrm ? s[i] / nrm[(long)n * P + p] : s[i];
    __syncthreads();
    for (int i = threadIdx.x; i < CS; i += 256) {
      const int c = i % C0, d = i / C0;                     // destination order: d-major, c fastest
      dst[(((long)n * S + d) * P + p) * C0 + c] = row[c * S + d];
    }
  }
}

// The same permutation for the training step, both ways, FT pixels per workgroup so that global accesses on the volume side
// are FT * C0 * 4 contiguous bytes (the per-pixel kernel above writes 64-byte pieces when C0 = 16); the tile is
// [pixel][c][d] with d padded by one word: consecutive d on the 2-D side, consecutive (pixel, c) on the volume side, both
// conflict-free.  FOLD = false: src [N][P][C0*S] -> dst [N][S][P][C0];  FOLD = true: src [N][S][P][C0] -> dst [N][P][C0*S].
constexpr int FT = 4;
template <bool FOLD>
__global__ void __launch_bounds__(256) lift_permute_kernel(const float* __restrict__ src, float* __restrict__ dst, long P, int C0, int S) {
  extern __shared__ float tile[];                           // [FT][C0][S + 1]
  const int n = blockIdx.y, CS = C0 * S, SP = S + 1;
  const long p0 = (long)blockIdx.x * FT;
  const int npx = (int)min((long)FT, P - p0);
  const float* flat = (FOLD ? dst : src) + ((long)n * P + p0) * CS;           // 2-D side: [pixel][c*S + d]
  const float* vol = (FOLD ? src : dst) + (long)n * S * P * C0 + p0 * C0;     // volume side: + (d*P + px)*C0 + c
  if (!FOLD) {
    for (int i = threadIdx.x; i < npx * CS; i += 256) {
      const int px = i / CS, j = i - px * CS, c = j / S, d = j - c * S;
      tile[(px * C0 + c) * SP + d] = flat[i];
    }
  } else {
    for (int i = threadIdx.x; i < S * npx * C0; i += 256) {
      const int c = i % C0, px = (i / C0) % npx, d = i / (C0 * npx);
      tile[(px * C0 + c) * SP + d] = vol[((long)d * P + px) * C0 + c];
    }
  }
  __syncthreads();
  if (!FOLD) {
    float* v = const_cast<float*>(vol);
    for (int i = threadIdx.x; i < S * npx * C0; i += 256) {
      const int c = i % C0, px = (i / C0) % npx, d = i / (C0 * npx);
      v[((long)d * P + px) * C0 + c] = tile[(px * C0 + c) * SP + d];
    }
  } else {
    float* f = const_cast<float*>(flat);
    for (int i = threadIdx.x; i < npx * CS; i += 256) {
      const int px = i / CS, j = i - px * CS, c = j / S, d = j - c * S;
      f[i] = tile[(px * C0 + c) * SP + d];
    }
  }
}

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
// (the fused lift kernels below use ld4 / bf16x4p, declared with the 16-channel epilogue kernel further down)
template <int MODE>
int launch_rows(const float* a, const float* b, const float* nrm_in, float* out, float* nrm_out,
                long rows, int C, unsigned flags, float slope, float eps, hipStream_t s) {
  if (rows <= 0 || C <= 0) return LF_EINVAL;
  const bool vec = (C % 4 == 0) && pow2(C / 4) && (C / 4) <= 64 && lf_aligned16(a) && lf_aligned16(out) &&
                   (MODE == 0 || lf_aligned16(b));
  if (vec) {
    const int lpr = C / 4, rpb = 256 / lpr;
    const long blocks = (rows + rpb - 1) / rpb;
    const unsigned grid = (unsigned)(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL((rows_vec4_kernel<MODE>), dim3(grid), dim3(256), 0, s, a, b, nrm_in, out, nrm_out, rows, C, lpr,
                       flags, slope, eps);
  } else {
    const long blocks = (rows + 3) / 4;
    const unsigned grid = (unsigned)(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL((rows_wave_kernel<MODE>), dim3(grid), dim3(256), 0, s, a, b, nrm_in, out, nrm_out, rows, C,
                       flags, slope, eps);
  }
  return lf_launch_status();
}


// ---- epilogue backward of a 16-channel layer for the training step (round 5): gp = LeakyReLU'(y) PixelNorm'(gy; y, norm), with
// the bias gradient (column sums of the un-rounded gp) folded in, and the three arrays in fp32 or bf16 storage (IO bit 0: gy,
// bit 1: y, bit 2: gp).  A lane quad owns a row (voxel); block partial sums of the 16 columns go to `partial` and are reduced
// in a fixed order by colsum_final_kernel (fp64) -- no pass over gp just for the bias.
typedef __bf16 bf16x4p __attribute__((ext_vector_type(4)));
template <bool B16>
__device__ __forceinline__ f32x4 ld4(const void* p, long i) {
  if constexpr (B16) return __builtin_convertvector(((const bf16x4p*)p)[i], f32x4);
  else return ((const f32x4*)p)[i];
}
template <int IO>
__global__ void __launch_bounds__(256) epilogue_bwd_c16_kernel(const void* __restrict__ gy, const void* __restrict__ y,
                                                               const float* __restrict__ nrm, void* __restrict__ gp,
                                                               float* __restrict__ partial, long rows, int chunk, unsigned flags, float slope) {
  const int t = threadIdx.x, q = t & 3, slot = t >> 2;
  const long r0 = (long)blockIdx.x * chunk, r1 = min(r0 + chunk, rows);
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long rb = r0; rb < r1; rb += 64) {
    const long row = rb + slot;
    const bool live = row < r1;
    f32x4 va = (f32x4){0.f, 0.f, 0.f, 0.f}, vb = va;
    if (live) {
      va = ld4<(IO & 1) != 0>(gy, row * 4 + q);
      if (flags != 0) vb = ld4<(IO & 2) != 0>(y, row * 4 + q);      // (flags = 0: y may be NULL / of another storage type)
    }
    f32x4 g = va;
    if (flags & LF_EPI_PIXELNORM) {
      float dot = va[0] * vb[0] + va[1] * vb[1] + va[2] * vb[2] + va[3] * vb[3];
      dot = group_sum(dot, 4) / 16.f;
      const float r = live ? nrm[row] : 1.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (va[e] - vb[e] * dot) / r;
    }
    if (flags & LF_EPI_LRELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = vb[e] > 0.f ? g[e] : g[e] * slope;
    }
    if (live) {
      if constexpr ((IO & 4) != 0) ((bf16x4p*)gp)[row * 4 + q] = __builtin_convertvector(g, bf16x4p);
      else ((f32x4*)gp)[row * 4 + q] = g;
      acc += g;
    }
  }
  if (partial == nullptr) return;
  __shared__ f32x4 red[256];
  red[t] = acc;
  __syncthreads();
  if (t < 16) {
    float sum = 0.f;
    for (int i = 0; i < 64; ++i) sum += red[i * 4 + (t >> 2)][t & 3];
    partial[(long)blockIdx.x * 16 + t] = sum;
  }
}

__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
  const int t = threadIdx.x, co = t & 15, grp = t >> 4;
  double sum = 0.0;
  for (int b = grp; b < nblk; b += 16) sum += (double)partial[(long)b * 16 + co];
  __shared__ double red[256];
  red[t] = sum;
  __syncthreads();
  if (t < 16) {
    double tot = 0.0;
    for (int g = 0; g < 16; ++g) tot += red[g * 16 + t];
    out[t] = (float)tot;
  }
}

// ---- FactorProjection2d3d for the training step, fused (round 5): the pointwise convolution leaves u = LeakyReLU(conv * he + b)
// as [N][P][C0*S] rows (channel = c*S + d); ONE pass then forms the PixelNorm over all C0*S channels of a pixel and writes the
// normalised values straight into the (N,C0,S,H,W) channels-last volume (fp32 or bf16 storage) + one norm per pixel -- instead
// of a normalisation pass and a permutation pass over 4 GB each.  The backward pass reads the volume-layout gradient and the
// saved volume, forms PixelNorm' / LeakyReLU' per pixel and writes the pre-activation gradient as rows for the weight / data
// gradient GEMMs (optionally already rounded to bf16 values: the autocast policy's operand rounding).
// Same tile as lift_permute_kernel: FT pixels per workgroup, [pixel][c][d (+1 pad)] floats in LDS; global accesses are 16-byte
// vectors on the row side (4 consecutive d) and 4-channel groups on the volume side.
template <bool OUT16>
__global__ void __launch_bounds__(256) lift_norm_unfold_kernel(const float* __restrict__ src, void* __restrict__ dst,
                                                               float* __restrict__ norm_out, long P, int C0, int S, float eps) {
  extern __shared__ float tile[];                           // [FT][C0][S + 1], then FT * 4 partial sums
  const int n = blockIdx.y, CS = C0 * S, SP = S + 1, tid = threadIdx.x;
  const long p0 = (long)blockIdx.x * FT;
  const int npx = (int)min((long)FT, P - p0);
  const f32x4* flat = (const f32x4*)(src + ((long)n * P + p0) * CS);
  float* part = tile + FT * C0 * SP;                        // [FT][4 waves]
  float ss[FT];
#pragma unroll
  for (int k = 0; k < FT; ++k) ss[k] = 0.f;
  const int nq = npx * CS / 4;
  for (int i = tid; i < nq; i += 256) {
    const f32x4 v = flat[i];
    const int e = i * 4, px = e / CS, j = e - px * CS, c = j / S, d = j - c * S;
    float* t = tile + (px * C0 + c) * SP + d;
    t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
    const float q = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
#pragma unroll
    for (int k = 0; k < FT; ++k) ss[k] += (px == k) ? q : 0.f;
  }
#pragma unroll
  for (int k = 0; k < FT; ++k) {
    float v = ss[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((tid & 63) == 0) part[k * 4 + (tid >> 6)] = v;
  }
  __syncthreads();
  float rn[FT];
#pragma unroll
  for (int k = 0; k < FT; ++k) rn[k] = sqrtf(((part[k * 4] + part[k * 4 + 1]) + (part[k * 4 + 2] + part[k * 4 + 3])) / (float)CS + eps);
  if (tid < npx) norm_out[(long)n * P + p0 + tid] = rn[tid];
  const int C4 = C0 / 4;
  char* vol = (char*)dst + ((long)n * S * P * C0 + p0 * C0) * (OUT16 ? 2 : 4);
  for (int i = tid; i < S * npx * C4; i += 256) {
    const int c4 = i % C4, px = (i / C4) % npx, d = i / (C4 * npx);
    const float* t = tile + (px * C0 + c4 * 4) * SP + d;
    float r = rn[0];
#pragma unroll
    for (int k = 1; k < FT; ++k) r = (px == k) ? rn[k] : r;
    const f32x4 o = (f32x4){t[0] / r, t[SP] / r, t[2 * SP] / r, t[3 * SP] / r};
    const long off = ((long)d * P + px) * C0 + c4 * 4;
    if constexpr (OUT16) *(bf16x4p*)(vol + off * 2) = __builtin_convertvector(o, bf16x4p);
    else *(f32x4*)(vol + off * 4) = o;
  }
}

// IO bit 0: the gradient volume, bit 1: the saved volume are bf16.  round_bf16 != 0: gp is written as bf16 VALUES (fp32 container).
template <int IO>
__global__ void __launch_bounds__(256) lift_bwd_fused_kernel(const void* __restrict__ gvol, const void* __restrict__ yvol,
                                                             const float* __restrict__ nrm, float* __restrict__ gp, long P, int C0,
                                                             int S, float slope, int round_bf16) {
  extern __shared__ float tile[];                           // g: [FT][C0][S + 1]; y: the same; then FT * 4 partial sums
  const int n = blockIdx.y, CS = C0 * S, SP = S + 1, tid = threadIdx.x, C4 = C0 / 4;
  const long p0 = (long)blockIdx.x * FT;
  const int npx = (int)min((long)FT, P - p0);
  float* tg = tile;
  float* ty = tile + FT * C0 * SP;
  float* part = ty + FT * C0 * SP;
  const long vbase = (long)n * S * P * C0 + p0 * C0;
  float dot[FT];
#pragma unroll
  for (int k = 0; k < FT; ++k) dot[k] = 0.f;
  for (int i = tid; i < S * npx * C4; i += 256) {
    const int c4 = i % C4, px = (i / C4) % npx, d = i / (C4 * npx);
    const long off = vbase + ((long)d * P + px) * C0 + c4 * 4;
    const f32x4 g = ld4<(IO & 1) != 0>(gvol, off >> 2), y = ld4<(IO & 2) != 0>(yvol, off >> 2);
    const int t = (px * C0 + c4 * 4) * SP + d;
#pragma unroll
    for (int e = 0; e < 4; ++e) { tg[t + e * SP] = g[e]; ty[t + e * SP] = y[e]; }
    const float q = g[0] * y[0] + g[1] * y[1] + g[2] * y[2] + g[3] * y[3];
#pragma unroll
    for (int k = 0; k < FT; ++k) dot[k] += (px == k) ? q : 0.f;
  }
#pragma unroll
  for (int k = 0; k < FT; ++k) {
    float v = dot[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((tid & 63) == 0) part[k * 4 + (tid >> 6)] = v;
  }
  __syncthreads();
  float dt[FT], rn[FT];
#pragma unroll
  for (int k = 0; k < FT; ++k) {
    dt[k] = ((part[k * 4] + part[k * 4 + 1]) + (part[k * 4 + 2] + part[k * 4 + 3])) / (float)CS;
    rn[k] = k < npx ? nrm[(long)n * P + p0 + k] : 1.f;
  }
  f32x4* flat = (f32x4*)(gp + ((long)n * P + p0) * CS);
  const int nq = npx * CS / 4;
  for (int i = tid; i < nq; i += 256) {
    const int e0 = i * 4, px = e0 / CS, j = e0 - px * CS, c = j / S, d = j - c * S;
    const int t = (px * C0 + c) * SP + d;
    float dk = dt[0], rk = rn[0];
#pragma unroll
    for (int k = 1; k < FT; ++k) { dk = (px == k) ? dt[k] : dk; rk = (px == k) ? rn[k] : rk; }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float yv = ty[t + e];
      float v = (tg[t + e] - yv * dk) / rk;
      v = yv > 0.f ? v : v * slope;
      o[e] = round_bf16 ? (float)(__bf16)v : v;
    }
    flat[i] = o;
  }
}

}  // namespace

extern "C" int lf_pixelnorm_fwd(const float* x, float* y, float* norm_out, long rows, int C, float eps, void* stream) {
  lf_clear_error();
  return launch_rows<0>(x, nullptr, nullptr, y, norm_out, rows, C, LF_EPI_PIXELNORM, 0.f, eps, (hipStream_t)stream);
}

extern "C" int lf_epilogue_bwd(const float* gy, const float* y, const float* norm, float* gp,
                               long rows, int C, unsigned flags, float slope, void* stream) {
  lf_clear_error();
  if ((flags & LF_EPI_PIXELNORM) && norm == nullptr) return LF_EINVAL;
  return launch_rows<1>(gy, y, norm, gp, nullptr, rows, C, flags, slope, 0.f, (hipStream_t)stream);
}

extern "C" size_t lf_epilogue_bwd_c16_scratch_bytes(long rows) {
  if (rows <= 0) return 0;
  long chunk = (rows + 2047) / 2048;
  chunk = (chunk + 63) / 64 * 64;
  return (size_t)((rows + chunk - 1) / chunk) * 16 * sizeof(float);
}

extern "C" int lf_epilogue_bwd_c16(const void* gy, const void* y, const float* norm, void* gp, float* gbias, void* scratch,
                                   size_t scratch_bytes, long rows, unsigned flags, float slope, int io, void* stream) {
  lf_clear_error();
  if (rows <= 0 || io < 0 || io > 7 || gy == nullptr || gp == nullptr) return LF_EINVAL;
  if (flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) return LF_EINVAL;
  if ((flags & LF_EPI_PIXELNORM) && norm == nullptr) return LF_EINVAL;
  if (flags != 0 && y == nullptr) return LF_EINVAL;
  if (!lf_aligned16(gy) || !lf_aligned16(gp) || (y && !lf_aligned16(y))) return LF_EALIGN;
  long chunk = (rows + 2047) / 2048;                              // ~2048 blocks, rows in multiples of the 64 a block covers
  chunk = (chunk + 63) / 64 * 64;
  const int nblk = (int)((rows + chunk - 1) / chunk);
  if (gbias != nullptr && (scratch == nullptr || scratch_bytes < (size_t)nblk * 16 * sizeof(float))) return LF_ENOSPC;
  hipStream_t s = (hipStream_t)stream;
  typedef void (*kern_t)(const void*, const void*, const float*, void*, float*, long, int, unsigned, float);
  static const kern_t kerns[8] = {epilogue_bwd_c16_kernel<0>, epilogue_bwd_c16_kernel<1>, epilogue_bwd_c16_kernel<2>, epilogue_bwd_c16_kernel<3>,
                                  epilogue_bwd_c16_kernel<4>, epilogue_bwd_c16_kernel<5>, epilogue_bwd_c16_kernel<6>, epilogue_bwd_c16_kernel<7>};
  hipLaunchKernelGGL(kerns[io], dim3(nblk), dim3(256), 0, s, gy, y ? y : gy, norm, gp, gbias ? (float*)scratch : nullptr, rows,
                     (int)chunk, flags, slope);
  int st = lf_launch_status();
  if (st || gbias == nullptr) return st;
  hipLaunchKernelGGL(colsum_final_kernel, dim3(1), dim3(256), 0, s, (const float*)scratch, nblk, gbias);
  return lf_launch_status();
}

extern "C" int lf_nchw_to_nhwc(const float* src, float* dst, int N, int C, long P, void* stream) {
  lf_clear_error();
  if (N <= 0 || C <= 0 || P <= 0) return LF_EINVAL;
  dim3 grid((unsigned)((P + 31) / 32), (unsigned)((C + 31) / 32), N);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, C, P);
  return lf_launch_status();
}

extern "C" int lf_nhwc_to_nchw(const float* src, float* dst, int N, int C, long P, void* stream) {
  lf_clear_error();
  if (N <= 0 || C <= 0 || P <= 0) return LF_EINVAL;
  if (P > 0x7fffffffL) return LF_EINVAL;
  // src [n][P][C] -> dst [n][C][P]: the same transpose with R = P, Q = C
  dim3 grid((unsigned)((C + 31) / 32), (unsigned)((P + 31) / 32), N);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, src, dst, (int)P, (long)C);
  return lf_launch_status();
}

extern "C" int lf_lift_unfold(const float* src, const float* norm_or_null, float* dst,
                              int N, long P, int C0, int S, void* stream) {
  lf_clear_error();
  if (N <= 0 || P <= 0 || C0 <= 0 || S <= 0) return LF_EINVAL;
  const size_t shmem = (size_t)C0 * S * sizeof(float);
  if (shmem > 64 * 1024) return LF_EINVAL;
  const unsigned gx = (unsigned)(P < 8192 ? P : 8192);
  hipLaunchKernelGGL(lift_unfold_kernel, dim3(gx, N), dim3(256), shmem, (hipStream_t)stream, src, norm_or_null, dst, P, C0, S);
  return lf_launch_status();
}

extern "C" int lf_lift_permute(const float* src, float* dst, int N, long P, int C0, int S, int fold, void* stream) {
  lf_clear_error();
  if (N <= 0 || P <= 0 || C0 <= 0 || S <= 0 || src == nullptr || dst == nullptr) return LF_EINVAL;
  const size_t shmem = (size_t)FT * C0 * (S + 1) * sizeof(float);
  if (shmem > 64 * 1024 || (P + FT - 1) / FT > 0x7fffffffL) return LF_EINVAL;
  const dim3 grid((unsigned)((P + FT - 1) / FT), (unsigned)N);
  if (fold)
    hipLaunchKernelGGL((lift_permute_kernel<true>), grid, dim3(256), shmem, (hipStream_t)stream, src, dst, P, C0, S);
  else
    hipLaunchKernelGGL((lift_permute_kernel<false>), grid, dim3(256), shmem, (hipStream_t)stream, src, dst, P, C0, S);
  return lf_launch_status();
}

static bool lift_fused_ok(int N, long P, int C0, int S) {
  return N > 0 && P > 0 && C0 > 0 && S > 0 && C0 % 4 == 0 && S % 4 == 0 && N <= 65535 && (P + FT - 1) / FT <= 0x7fffffffL &&
         (size_t)(2 * FT * C0 * (S + 1) + 4 * FT) * sizeof(float) <= 150 * 1024;
}

extern "C" int lf_lift_norm_unfold(const float* src, void* dst, float* norm_out, int N, long P, int C0, int S, float eps,
                                   int out_bf16, void* stream) {
  lf_clear_error();
  if (!lift_fused_ok(N, P, C0, S) || src == nullptr || dst == nullptr || norm_out == nullptr) return LF_EINVAL;
  if (!lf_aligned16(src) || !lf_aligned16(dst)) return LF_EALIGN;
  const size_t shmem = (size_t)(FT * C0 * (S + 1) + 4 * FT) * sizeof(float);
  const dim3 grid((unsigned)((P + FT - 1) / FT), (unsigned)N);
  typedef void (*kern_t)(const float*, void*, float*, long, int, int, float);
  const kern_t k = out_bf16 ? (kern_t)lift_norm_unfold_kernel<true> : (kern_t)lift_norm_unfold_kernel<false>;
  static lf_devmask_t attr[2];
  if (shmem > 48 * 1024) {
    hipError_t e = lf_ensure_dyn_lds(attr[out_bf16 ? 1 : 0], (const void*)k, 150 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), shmem, (hipStream_t)stream, src, dst, norm_out, P, C0, S, eps);
  return lf_launch_status();
}

extern "C" int lf_lift_bwd(const void* gvol, const void* yvol, const float* norm, float* gp, int N, long P, int C0, int S,
                           float slope, int round_bf16, int io, void* stream) {
  lf_clear_error();
  if (!lift_fused_ok(N, P, C0, S) || gvol == nullptr || yvol == nullptr || norm == nullptr || gp == nullptr || io < 0 || io > 3)
    return LF_EINVAL;
  if (!lf_aligned16(gvol) || !lf_aligned16(yvol) || !lf_aligned16(gp)) return LF_EALIGN;
  const size_t shmem = (size_t)(2 * FT * C0 * (S + 1) + 4 * FT) * sizeof(float);
  const dim3 grid((unsigned)((P + FT - 1) / FT), (unsigned)N);
  typedef void (*kern_t)(const void*, const void*, const float*, float*, long, int, int, float, int);
  static const kern_t kerns[4] = {lift_bwd_fused_kernel<0>, lift_bwd_fused_kernel<1>, lift_bwd_fused_kernel<2>, lift_bwd_fused_kernel<3>};
  static lf_devmask_t attr[4];
  if (shmem > 48 * 1024) {
    hipError_t e = lf_ensure_dyn_lds(attr[io], (const void*)kerns[io], 150 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kerns[io], grid, dim3(256), shmem, (hipStream_t)stream, gvol, yvol, norm, gp, P, C0, S, slope, round_bf16);
  return lf_launch_status();
}

extern "C" int lf_abi_version(void) {
  lf_clear_error(); return LF_ABI_VERSION; }

extern "C" int lf_device_name(char* buf, int buflen) {
  lf_clear_error();
  hipDeviceProp_t prop;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return (int)e;
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}
