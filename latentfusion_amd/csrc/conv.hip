// He-equalised convolutions with fused epilogue on the gfx950 matrix cores (exact-fp32 MFMA).
//
//   y = conv(x, W) * he + bias ; LeakyReLU ; PixelNorm          (one half of Block.forward,
//   latentfusion/modules/blocks.py:152-158 + equalized.py:57-64 + modules/__init__.py:14-15)
//
// Implicit GEMM, channels-last.  One MFMA v_mfma_f32_16x16x4_f32 computes a [16 cout] x [16 voxel]
// tile over 4 input channels:
//   A (weights)  lane l holds W[cout = l&15][cin = 4*(l>>4) + s]          (s = step 0..3)
//   B (inputs)   lane l holds X[voxel = l&15][cin = 4*(l>>4) + s]
//   D            lane l holds Y[cout = 4*(l>>4) + e][voxel = l&15], e = 0..3
// so one 16-byte load per lane feeds four MFMAs (K = 16 channels) and every lane ends up with
// four CONSECUTIVE output channels of one voxel: the epilogue stores float4 and the PixelNorm
// channel reduction is two xor-shuffles (lanes l, l^16, l^32, l^48 share a voxel).
//
// 3x3(x3) kernel: a 256-thread workgroup owns a 4x4x16 (3-D) or 16x16 (2-D) output tile, stages
// the zero-padded halo tile of 16 input channels in LDS (41.5 KB / 20.7 KB), each wave computes
// four 16-voxel rows so that every weight fragment is reused four times from registers.
#include "lf_common.h"

namespace {

constexpr int TX = 16;

template <int DIMS> struct TileGeom;
template <> struct TileGeom<3> { static constexpr int TY = 4, TZ = 4, HY = 6, HZ = 6, TAPS = 27; };
template <> struct TileGeom<2> { static constexpr int TY = 16, TZ = 1, HY = 18, HZ = 1, TAPS = 9; };

__device__ __forceinline__ f32x4 mfma4(const f32x4 a, const f32x4 b, f32x4 c) {
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], c, 0, 0, 0);
  return c;
}

// Shared epilogue: acc[t][j] = raw conv sums of cout tile t for voxel-row j.
// rowoff[j] = float offset of the row's output record in y (< 0: lane/row outside the tensor),
// rowidx[j] = flat voxel index (for norm_out).  Output channel co lands at
// rowoff + (co / ysc) * yss + (co % ysc): ysc >= Cout gives the plain channels-last record,
// smaller ysc scatters channel slices (depth-unfolded outputs).
template <int NT, int NR>
__device__ __forceinline__ void epilogue_store(f32x4 (&acc)[NT][NR], const long (&rowoff)[NR], const long (&rowidx)[NR],
                                               const float* __restrict__ bias, float* __restrict__ y,
                                               float* __restrict__ norm_out, int Cout, int co_base,
                                               int ysc, long yss, bool vec_out,
                                               float he, unsigned flags, float slope, float eps) {
  const int lane = threadIdx.x & 63;
  const int cq = lane >> 4;
  float bv[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int co = co_base + t * 16 + cq * 4 + e;
      bv[t][e] = (bias != nullptr && co < Cout) ? bias[co] : 0.f;
    }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    float ss = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co_base + t * 16 + cq * 4 + e;
        float v = acc[t][j][e] * he + bv[t][e];
        if (flags & LF_EPI_LRELU) v = lf_lrelu(v, slope);
        if (co >= Cout) v = 0.f;
        acc[t][j][e] = v;
        ss += v * v;
      }
    float r = 1.f;
    if (flags & LF_EPI_PIXELNORM) {
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      r = sqrtf(ss / (float)Cout + eps);
    }
    if (rowoff[j] >= 0) {
      float* dst = y + rowoff[j];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int co = co_base + t * 16 + cq * 4;
        f32x4 v = acc[t][j];
        if (flags & LF_EPI_PIXELNORM) { v[0] /= r; v[1] /= r; v[2] /= r; v[3] /= r; }
        if (vec_out) {
          if (co < Cout) *(f32x4*)(dst + (long)(co / ysc) * yss + (co % ysc)) = v;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (co + e < Cout) dst[(long)((co + e) / ysc) * yss + ((co + e) % ysc)] = v[e];
        }
      }
      if ((flags & LF_EPI_PIXELNORM) && norm_out != nullptr && cq == 0) norm_out[rowidx[j]] = r;
    }
  }
}

template <int DIMS, int NT>
__global__ void __launch_bounds__(256) conv3x3_kernel(
    const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int N, int D, int H, int W, int Cin, int Cout, int CinP, int CoutP,
    int tiles_x, int tiles_y, int tiles_z,
    float he, unsigned flags, float slope, float eps) {
  using G = TileGeom<DIMS>;
  constexpr int HX = TX + 2;
  constexpr int HALO = G::HZ * G::HY * HX;
  __shared__ __attribute__((aligned(16))) float tile[HALO * 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, cq = lane >> 4;

  int b = blockIdx.x;
  const int bx = b % tiles_x; b /= tiles_x;
  const int by = b % tiles_y; b /= tiles_y;
  const int bz = b % tiles_z; b /= tiles_z;
  const int n = b;
  const int x0 = bx * TX, y0 = by * G::TY, z0 = bz * G::TZ;
  const int co_base = blockIdx.y * NT * 16;

  f32x4 acc[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // LDS float offset of (row j of this wave, halo origin) for this lane
  int rowbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 4 + j;
    const int ry = (DIMS == 3) ? (r & 3) : r;
    const int rz = (DIMS == 3) ? (r >> 2) : 0;
    rowbase[j] = ((rz * G::HY + ry) * HX + li) * 16 + cq * 4;
  }

  const bool vec_in = ((Cin & 3) == 0);
  const int nchunks = CinP >> 4;
  for (int ch = 0; ch < nchunks; ++ch) {
    if (ch) __syncthreads();
    // ---- stage the zero-padded halo tile of channels [16*ch, 16*ch+16) ----
    for (int i = tid; i < HALO * 4; i += 256) {
      const int q = i & 3;
      int v = i >> 2;
      const int lx = v % HX; v /= HX;
      const int ly = v % G::HY;
      const int lz = v / G::HY;
      const int gx = x0 + lx - 1, gy = y0 + ly - 1;
      const int gz = (DIMS == 3) ? (z0 + lz - 1) : 0;
      f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int c0 = ch * 16 + q * 4;
      if (gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D && c0 < Cin) {
        const float* src = x + ((((long)n * D + gz) * H + gy) * W + gx) * Cin + c0;
        if (vec_in) {
          val = *(const f32x4*)src;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (c0 + e < Cin) val[e] = src[e];
        }
      }
      *(f32x4*)(tile + (i >> 2) * 16 + q * 4) = val;
    }
    __syncthreads();
    // ---- 27 (9) taps x NT cout tiles x 4 rows ----
    const float* wch = wpack + (long)(co_base + li) * CinP + ch * 16 + cq * 4;
#pragma unroll 1
    for (int tap = 0; tap < G::TAPS; ++tap) {
      const int kx = tap % 3;
      const int ky = (tap / 3) % 3;
      const int kz = (DIMS == 3) ? tap / 9 : 0;
      const int toff = ((kz * G::HY + ky) * HX + kx) * 16;
      f32x4 bfrag[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfrag[j] = *(const f32x4*)(tile + rowbase[j] + toff);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 afrag = *(const f32x4*)(wch + ((long)tap * CoutP + t * 16) * CinP);
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
          for (int j = 0; j < 4; ++j)   // four independent accumulators hide the 40-cycle MFMA latency
            acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[st], bfrag[j][st], acc[t][j], 0, 0, 0);
      }
    }
  }

  long rowoff[4], rowidx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 4 + j;
    const int gy = y0 + ((DIMS == 3) ? (r & 3) : r);
    const int gz = (DIMS == 3) ? z0 + (r >> 2) : 0;
    const int gx = x0 + li;
    rowidx[j] = (gx < W && gy < H && gz < D) ? ((((long)n * D + gz) * H + gy) * W + gx) : -1;
    rowoff[j] = rowidx[j] < 0 ? -1 : rowidx[j] * Cout;
  }
  epilogue_store<NT, 4>(acc, rowoff, rowidx, bias, y, norm_out, Cout, co_base, 1 << 30, 0, (Cout & 3) == 0,
                        he, flags, slope, eps);
}

// ---- pointwise convolution: GEMM over pixels, depth axis optionally folded into K ------------
template <int NT>
__global__ void __launch_bounds__(256) conv1x1_kernel(
    const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int P, int Cin, int ksl, long x_batch_stride, long x_slice_stride, int Cout, int Kp,
    long y_batch_stride, int y_row_stride, int y_slice_channels, long y_slice_stride,
    float he, unsigned flags, float slope, float eps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, cq = lane >> 4;
  const int n = blockIdx.z;
  const int co_base = blockIdx.y * NT * 16;
  const long p = (long)blockIdx.x * 64 + wave * 16 + li;
  const bool live = p < P;
  const bool vec_in = ((Cin & 3) == 0);
  const float* xrow = x + (long)n * x_batch_stride + (live ? p : 0) * Cin;
  const float* wrow = wpack + (long)(co_base + li) * Kp + cq * 4;

  f32x4 acc[NT][1];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int K = ksl * Cin;
  const int nchunks = Kp >> 4;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int k0 = kc * 16 + cq * 4;
    f32x4 bfrag = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live && k0 < K) {
      const int s = k0 / Cin, c0 = k0 - s * Cin;      // ksl > 1 implies Cin % 4 == 0: no straddling
      const float* src = xrow + (long)s * x_slice_stride + c0;
      if (vec_in) {
        bfrag = *(const f32x4*)src;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k0 + e < K) bfrag[e] = src[e];
      }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 afrag = *(const f32x4*)(wrow + (long)t * 16 * Kp + kc * 16);
      acc[t][0] = mfma4(afrag, bfrag, acc[t][0]);
    }
  }
  long rowidx[1] = {live ? ((long)n * P + p) : -1};
  long rowoff[1] = {live ? ((long)n * y_batch_stride + p * y_row_stride) : -1};
  const bool vec_out = ((y_row_stride | y_slice_channels) & 3) == 0 && ((y_batch_stride | y_slice_stride) & 3) == 0;
  epilogue_store<NT, 1>(acc, rowoff, rowidx, bias, y, norm_out, Cout, co_base, y_slice_channels, y_slice_stride,
                        vec_out, he, flags, slope, eps);
}

}  // namespace

extern "C" int lf_conv3x3_cout_padded(int Cout) {
  const int c16 = (Cout + 15) & ~15;
  if (c16 <= 16) return 16;
  if (c16 <= 32) return 32;
  return (c16 + 63) & ~63;
}

extern "C" int lf_conv1x1_cout_padded(int Cout) {
  const int c16 = (Cout + 15) & ~15;
  if (c16 <= 16) return 16;
  if (c16 <= 32) return 32;
  if (c16 <= 64) return 64;
  return (c16 + 127) & ~127;
}

extern "C" int lf_conv3x3_fwd(const float* x, const float* wpack, const float* bias, float* y, float* norm_out,
                              int dims, int N, int D, int H, int W, int Cin, int Cout,
                              float he, unsigned flags, float slope, float eps, void* stream) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return LF_EINVAL;
  if (dims != 2 && dims != 3) return LF_EINVAL;
  if (dims == 2 && D != 1) return LF_EINVAL;
  if ((flags & LF_EPI_PIXELNORM) && Cout > 64) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(y) || !lf_aligned16(wpack)) return LF_EALIGN;
  const int CinP = (Cin + 15) & ~15, CoutP = lf_conv3x3_cout_padded(Cout);
  const int ntiles = CoutP / 16;
  const int NT = ntiles >= 4 ? 4 : ntiles;          // 1, 2 or 4 (CoutP is padded accordingly)
  const int groups = ntiles / NT;
  const int TY = dims == 3 ? 4 : 16, TZ = dims == 3 ? 4 : 1;
  const int tiles_x = (W + TX - 1) / TX, tiles_y = (H + TY - 1) / TY, tiles_z = (D + TZ - 1) / TZ;
  const long nblk = (long)tiles_x * tiles_y * tiles_z * N;
  if (nblk > 0x7fffffffL) return LF_EINVAL;
  dim3 grid((unsigned)nblk, groups), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(DM, T) hipLaunchKernelGGL((conv3x3_kernel<DM, T>), grid, block, 0, s, x, wpack, bias, y, norm_out, \
                                         N, D, H, W, Cin, Cout, CinP, CoutP, tiles_x, tiles_y, tiles_z, he, flags, slope, eps)
  if (dims == 3) { if (NT == 4) LAUNCH(3, 4); else if (NT == 2) LAUNCH(3, 2); else LAUNCH(3, 1); }
  else           { if (NT == 4) LAUNCH(2, 4); else if (NT == 2) LAUNCH(2, 2); else LAUNCH(2, 1); }
#undef LAUNCH
  return lf_launch_status();
}

extern "C" int lf_conv1x1_fwd(const float* x, const float* wpack, const float* bias, float* y, float* norm_out,
                              int N, int P, int Cin, int ksl, long x_batch_stride, long x_slice_stride,
                              int Cout, long y_batch_stride, int y_row_stride, int y_slice_channels,
                              long y_slice_stride,
                              float he, unsigned flags, float slope, float eps, void* stream) {
  if (N <= 0 || P <= 0 || Cin <= 0 || ksl <= 0 || Cout <= 0 || y_slice_channels <= 0) return LF_EINVAL;
  if (y_row_stride < (y_slice_channels < Cout ? y_slice_channels : Cout)) return LF_EINVAL;
  if (ksl > 1 && (Cin & 3)) return LF_EALIGN;     // a lane's 4-channel group must not straddle slices
  if ((flags & LF_EPI_PIXELNORM) && Cout > 128) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(y) || !lf_aligned16(wpack)) return LF_EALIGN;
  if ((Cin & 3) == 0 && ((x_batch_stride | x_slice_stride) & 3)) return LF_EALIGN;
  const int K = ksl * Cin, Kp = (K + 15) & ~15;
  const int CoutP = lf_conv1x1_cout_padded(Cout);
  const int ntiles = CoutP / 16;
  const int NT = ntiles >= 8 ? 8 : ntiles;          // 1, 2, 4 or 8 (CoutP is padded accordingly)
  const int groups = ntiles / NT;
  dim3 grid((unsigned)((P + 63) / 64), groups, N), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T) hipLaunchKernelGGL((conv1x1_kernel<T>), grid, block, 0, s, x, wpack, bias, y, norm_out, \
                                     P, Cin, ksl, x_batch_stride, x_slice_stride, Cout, Kp, y_batch_stride, y_row_stride, \
                                     y_slice_channels, y_slice_stride, he, flags, slope, eps)
  switch (NT) {
    case 1: LAUNCH(1); break;
    case 2: LAUNCH(2); break;
    case 4: LAUNCH(4); break;
    case 8: LAUNCH(8); break;
    default: return LF_EINVAL;
  }
#undef LAUNCH
  return lf_launch_status();
}
