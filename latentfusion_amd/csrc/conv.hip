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

// Shared epilogue: acc[t][j] = raw conv sums of cout tile t for voxel-row j.
// rowoff[j] = float offset of the row's output record in y (< 0: lane/row outside the tensor),
// rowidx[j] = flat voxel index (for norm_out).  Output channel co lands at
// rowoff + (co / ysc) * yss + (co % ysc): ysc >= Cout gives the plain channels-last record,
// smaller ysc scatters channel slices (depth-unfolded outputs).
template <int NT, int NR, bool WHOLE = false>
__device__ __forceinline__ void epilogue_store(f32x4 (&acc)[NT][NR], const long (&rowoff)[NR], const long (&rowidx)[NR],
                                               const float* __restrict__ bias, float* __restrict__ y,
                                               float* __restrict__ norm_out, int Cout, int co_base,
                                               int ysc, long yss, bool vec_out,
                                               float he, unsigned flags, float slope, float eps,
                                               const float* __restrict__ prev_y = nullptr,
                                               const float* __restrict__ prev_norm = nullptr, unsigned prev_flags = 0,
                                               float* __restrict__ amax_out = nullptr) {
  float lane_amax = 0.f;
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
    float r = 1.f, rinv = 1.f;
    if (flags & LF_EPI_PIXELNORM) {
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      r = sqrtf(ss / (float)Cout + eps);
      rinv = 1.0f / r;          // one IEEE division per voxel; y = v * (1/r) is within 1 ulp of v / r
    }
    if (prev_y != nullptr) {
      // Data-gradient use: acc holds dL/d(output of the previous layer's epilogue) for 16-channel
      // records; fold that layer's LeakyReLU' * PixelNorm' in here (saves one HBM round trip):
      //   g <- lrelu'(y_prev) * (g - y_prev * mean_c(g * y_prev)) / norm_prev      (C == 16 per tile)
      // All NT loads are issued before the first use so they overlap each other.
      // Two record shapes: 16-channel slices (ysc == 16: the unfolding factor projection, every tile its own voxel and norm)
      // and -- round 4 -- WHOLE rows of Cout channels (ysc >= Cout, all of them in this workgroup's NT tiles: the 2-D decoder's
      // 32-channel layers), whose PixelNorm' dot product runs over all tiles of the row and whose norm is indexed by the row.
      // (compile-time: the multi-tile 3x3 kernels only -- the sliced pointwise kernel must not pay registers for it, and a
      // single 16-channel tile is the same arithmetic either way)
      constexpr bool whole = WHOLE;
      f32x4 ypv[NT];
      float nrv[NT];
      if (!whole && (prev_flags & LF_EPI_DOT)) {
        // (round 6, lf_conv1x1_bwd_data with LF_EPI_DOT) the gradient is stored as it is; on the side, per 16-channel record,
        // dot[record] = sum_c g[c] * prev_y[c] goes to the buffer passed as prev_norm: the gradient w.r.t. a per-voxel FACTOR
        // that scaled prev_y ahead of this convolution (the occlusion weights ahead of the factor projection) without a pass
        // over the two volumes (was lf_column_scale_bwd)
        float* dot_out = const_cast<float*>(prev_norm);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int co = co_base + t * 16 + cq * 4;
          const bool ok = rowoff[j] >= 0 && co < Cout;
          const long off = ok ? rowoff[j] + (long)(co / ysc) * yss + (co % ysc) : 0;
          ypv[t] = ok ? *(const f32x4*)(prev_y + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int co = co_base + t * 16 + cq * 4;
          const bool ok = rowoff[j] >= 0 && co < Cout;
          const long off = ok ? rowoff[j] + (long)(co / ysc) * yss + (co % ysc) : 0;
          const f32x4 g = acc[t][j], yp = ypv[t];
          float dot = g[0] * yp[0] + g[1] * yp[1] + g[2] * yp[2] + g[3] * yp[3];
          dot += __shfl_xor(dot, 16, 64);
          dot += __shfl_xor(dot, 32, 64);
          if (ok && cq == 0) dot_out[off >> 4] = dot;
        }
      } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int co = co_base + t * 16 + cq * 4;
        const bool ok = rowoff[j] >= 0 && co < Cout;
        const long off = ok ? rowoff[j] + (long)(co / ysc) * yss + (co % ysc) : 0;
        ypv[t] = ok ? *(const f32x4*)(prev_y + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
        nrv[t] = (ok && (prev_flags & LF_EPI_PIXELNORM)) ? prev_norm[whole ? rowidx[j] : (off >> 4)] : 1.f;
      }
      float dot_row = 0.f;
      if (whole && (prev_flags & LF_EPI_PIXELNORM)) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          dot_row += acc[t][j][0] * ypv[t][0] + acc[t][j][1] * ypv[t][1] + acc[t][j][2] * ypv[t][2] + acc[t][j][3] * ypv[t][3];
        dot_row += __shfl_xor(dot_row, 16, 64);
        dot_row += __shfl_xor(dot_row, 32, 64);
        dot_row *= 1.f / (float)Cout;
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const f32x4 yp = ypv[t];
        f32x4 g = acc[t][j];
        if (prev_flags & LF_EPI_PIXELNORM) {
          float dot = dot_row;
          if (!whole) {
            dot = g[0] * yp[0] + g[1] * yp[1] + g[2] * yp[2] + g[3] * yp[3];
            dot += __shfl_xor(dot, 16, 64);
            dot += __shfl_xor(dot, 32, 64);
            dot *= (1.f / 16.f);
          }
          const float rinv = 1.0f / nrv[t];
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = (g[e] - yp[e] * dot) * rinv;
        }
        if (prev_flags & LF_EPI_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = yp[e] > 0.f ? g[e] : g[e] * slope;
        }
        acc[t][j] = g;
        if (amax_out != nullptr && rowoff[j] >= 0)
          lane_amax = fmaxf(lane_amax, fmaxf(fmaxf(fabsf(g[0]), fabsf(g[1])), fmaxf(fabsf(g[2]), fabsf(g[3]))));
      }
      }
    }
    if (rowoff[j] >= 0 && y != nullptr) {                          // (y == NULL: LF_EPI_DOT launches that only want the sums)
      float* dst = y + rowoff[j];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int co = co_base + t * 16 + cq * 4;
        f32x4 v = acc[t][j];
        if (flags & LF_EPI_PIXELNORM) { v[0] *= rinv; v[1] *= rinv; v[2] *= rinv; v[3] *= rinv; }
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
  if (amax_out != nullptr) {                       // max-abs of what this wave wrote (order-independent)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lane_amax = fmaxf(lane_amax, __shfl_xor(lane_amax, o, 64));
    lf_amax_publish(amax_out, lane_amax, lane);
  }
}

// Stages the zero-padded halo tile of 16 input channels [c_first, c_first+16) into LDS.  All
// global loads of a thread (NST = ceil(HALO*4/256) float4s) are issued back to back into registers
// before the first LDS write, so one memory latency is paid per tile instead of one per load.
template <int DIMS>
__device__ __forceinline__ void stage_halo(float* __restrict__ tile, const float* __restrict__ x,
                                           int n, int x0, int y0, int z0, int D, int H, int W, int Cin,
                                           int c_first, bool vec_in) {
  using G = TileGeom<DIMS>;
  constexpr int HX = TX + 2;
  constexpr int HALO = G::HZ * G::HY * HX;
  constexpr int NST = (HALO * 4 + 255) / 256;
  const int tid = threadIdx.x;
  f32x4 st[NST];
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int i = tid + it * 256;
    const int q = i & 3;
    int v = i >> 2;
    const int lx = v % HX; v /= HX;
    const int ly = v % G::HY;
    const int lz = v / G::HY;
    const int gx = x0 + lx - 1, gy = y0 + ly - 1;
    const int gz = (DIMS == 3) ? (z0 + lz - 1) : 0;
    f32x4 val = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int c0 = c_first + q * 4;
    if (i < HALO * 4 && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D && c0 < Cin) {
      const float* src = x + ((((long)n * D + gz) * H + gy) * W + gx) * Cin + c0;
      if (vec_in) {
        val = *(const f32x4*)src;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (c0 + e < Cin) val[e] = src[e];
      }
    }
    st[it] = val;
  }
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int i = tid + it * 256;
    if (i < HALO * 4) *(f32x4*)(tile + i * 4) = st[it];
  }
}

template <int DIMS, int NT>
__global__ void __launch_bounds__(256) conv3x3_kernel(
    const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int N, int D, int H, int W, int Cin, int Cout, int CinP, int CoutP,
    int tiles_x, int tiles_y, int tiles_z,
    float he, unsigned flags, float slope, float eps,
    const float* __restrict__ prev_y, const float* __restrict__ prev_norm, unsigned prev_flags) {
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
    stage_halo<DIMS>(tile, x, n, x0, y0, z0, D, H, W, Cin, ch * 16, vec_in);
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
  epilogue_store<NT, 4, (NT > 1)>(acc, rowoff, rowidx, bias, y, norm_out, Cout, co_base, 1 << 30, 0, (Cout & 3) == 0,
                                  he, flags, slope, eps, prev_y, prev_norm, prev_flags);
}

// ---- C = 16 specialisation of the 3x3(x3) kernel (the SYN(S,16) hot path) ---------------------
// One Cin chunk, one Cout tile: all 27 (9) weight fragments live in registers (108 / 36 VGPRs), the
// tap loop is fully unrolled (constant LDS offsets) and the B fragments of tap t+1 are read from LDS
// before the 16 MFMAs of tap t are issued, so the matrix pipe never waits on LDS or VMEM.
template <int DIMS>
__global__ void __launch_bounds__(256, 2) conv3x3_c16_kernel(
    const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int N, int D, int H, int W, int Cin, int Cout,
    int tiles_x, int tiles_y, int tiles_z,
    float he, unsigned flags, float slope, float eps,
    const float* __restrict__ prev_y, const float* __restrict__ prev_norm, unsigned prev_flags) {
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

  // ---- weights -> registers (L2-resident 27 KB, read once per wave) ----
  f32x4 wreg[G::TAPS];
  {
    const float* wl = wpack + li * 16 + cq * 4;                 // [tap][16 cout][16 cin]
#pragma unroll
    for (int tap = 0; tap < G::TAPS; ++tap) wreg[tap] = *(const f32x4*)(wl + tap * 256);
  }

  // ---- stage the zero-padded halo tile ----
  stage_halo<DIMS>(tile, x, n, x0, y0, z0, D, H, W, Cin, 0, (Cin & 3) == 0);
  __syncthreads();

  const float* rowp[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 4 + j;
    const int ry = (DIMS == 3) ? (r & 3) : r;
    const int rz = (DIMS == 3) ? (r >> 2) : 0;
    rowp[j] = tile + ((rz * G::HY + ry) * HX + li) * 16 + cq * 4;
  }

  f32x4 acc[1][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[0][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 bcur[4], bnext[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) bcur[j] = *(const f32x4*)(rowp[j]);      // tap 0: offset 0
#pragma unroll
  for (int tap = 0; tap < G::TAPS; ++tap) {
    if (tap + 1 < G::TAPS) {
      constexpr int dummy = 0; (void)dummy;
      const int t1 = tap + 1;
      const int kx = t1 % 3, ky = (t1 / 3) % 3, kz = (DIMS == 3) ? t1 / 9 : 0;
      const int toff = ((kz * G::HY + ky) * HX + kx) * 16;
#pragma unroll
      for (int j = 0; j < 4; ++j) bnext[j] = *(const f32x4*)(rowp[j] + toff);
    }
#pragma unroll
    for (int st = 0; st < 4; ++st)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][st], bcur[j][st], acc[0][j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) bcur[j] = bnext[j];
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
  epilogue_store<1, 4>(acc, rowoff, rowidx, bias, y, norm_out, Cout, 0, 1 << 30, 0, (Cout & 3) == 0,
                       he, flags, slope, eps, prev_y, prev_norm, prev_flags);
}

// ---- persistent, double-buffered C = 16 conv3d kernel (the dominant kernel of the pose loop) ---
// One 512-thread workgroup per CU walks a contiguous run of 4x8x16 output tiles.
//  * weights stay in registers for the whole run (108 VGPRs, no per-tile L2 weight traffic);
//  * the zero-padded halo of tile t+1 (6x10x18 voxels x 16 ch = 67.5 KB) is DMA'd straight into
//    the other LDS buffer with `buffer_load_dwordx4 ... lds` while tile t is multiplied: no
//    staging registers, no ds_write, and out-of-volume voxels are zero-filled by the buffer
//    descriptor's range check (offset forced out of range) instead of by branches;
//  * 8 waves = 2 per SIMD, each wave owns four 16-voxel rows: 16 independent MFMAs per tap,
//    taps fully unrolled, B fragments of tap t+1 read from LDS before tap t's MFMAs issue.
namespace p3 {
constexpr int TZ = 4, TY = 8, HZ = 6, HY = 10, HX = TX + 2;
constexpr int HALO = HZ * HY * HX;                         // 1080 voxels
constexpr int NTHREADS = 512;
constexpr int NSLOT = (HALO * 4 + 63) / 64;                // 68 wave-wide 1 KiB DMA pieces per tile
constexpr int NIT = (NSLOT + 7) / 8;                       // 9 pieces per wave
constexpr int BUF_FLOATS = NSLOT * 256;                    // 69,632 B per buffer
}  // namespace p3

__global__ void __launch_bounds__(512, 2) conv3d_c16_persistent_kernel(
    const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z, int ntiles,
    float he, unsigned flags, float slope, float eps,
    const float* __restrict__ prev_y, const float* __restrict__ prev_norm, unsigned prev_flags) {
  __shared__ __attribute__((aligned(16))) float lds[2 * p3::BUF_FLOATS];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, cq = lane >> 4;

  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per;
  const int t_end = min(t_begin + per, ntiles);
  if (t_begin >= t_end) return;

  const long nvox = (long)D * H * W;
  const unsigned sample_bytes = (unsigned)(nvox * 64);

  // ---- per-lane constants of the DMA pieces this wave issues: piece s = wave + 8*it ----
  int rel[p3::NIT];        // byte offset of the lane's 16 B relative to the halo origin voxel
  int lxyz[p3::NIT];       // packed halo coordinates lx | ly << 8 | lz << 16 (0xffffff: beyond the halo)
#pragma unroll
  for (int it = 0; it < p3::NIT; ++it) {
    const int e = (wave + 8 * it) * 64 + lane;               // float4 index inside the halo buffer
    int v = e >> 2;
    const int q = e & 3;
    const int lx = v % p3::HX; v /= p3::HX;
    const int ly = v % p3::HY;
    const int lz = v / p3::HY;
    const bool inside = e < p3::HALO * 4;
    rel[it] = ((lz * H + ly) * W + lx) * 64 + q * 16;
    lxyz[it] = inside ? (lx | (ly << 8) | (lz << 16)) : 0x7f7f7f;
  }

  f32x4 wreg[27];
  {
    const float* wl = wpack + li * 16 + cq * 4;                 // [tap][16 cout][16 cin]
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) wreg[tap] = *(const f32x4*)(wl + tap * 256);
  }
  f32x4 bv4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) bv4 = *(const f32x4*)(bias + cq * 4);

  int rowoffs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 4 + j;                                  // rz = r / 8, ry = r % 8
    rowoffs[j] = (((r >> 3) * p3::HY + (r & 7)) * p3::HX + li) * 16 + cq * 4;
  }

  auto issue_dma = [&](int t, int buf) {
    int tt = t;
    const int bx = tt % tiles_x; tt /= tiles_x;
    const int by = tt % tiles_y; tt /= tiles_y;
    const int bz = tt % tiles_z; tt /= tiles_z;
    const int n = tt;
    const int ox = bx * TX - 1, oy = by * p3::TY - 1, oz = bz * p3::TZ - 1;      // halo origin (may be -1)
    const int tile_off = ((oz * H + oy) * W + ox) * 64;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)n * nvox * 16), 0,
                                                                  sample_bytes, 0x00020000);
#pragma unroll
    for (int it = 0; it < p3::NIT; ++it) {
      const int s = wave + 8 * it;
      if (s < p3::NSLOT) {                                         // wave-uniform
        const int gx = ox + (lxyz[it] & 0xff), gy = oy + ((lxyz[it] >> 8) & 0xff), gz = oz + (lxyz[it] >> 16);
        const bool ok = (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D;
        const int voff = ok ? (rel[it] + tile_off) : 0x7fffffff;   // out of range -> the DMA writes zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rs, (__attribute__((address_space(3))) void*)(lds + buf * p3::BUF_FLOATS + s * 256), 16, voff, 0, 0, 0);
      }
    }
  };

  issue_dma(t_begin, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x4 acc[4];

  // data-gradient launches: the previous layer's saved output / norm of the tile's 4 rows, fetched
  // BEFORE the tile's MFMAs so the global latency is hidden behind them
  f32x4 pyv[4];
  float pnv[4];
  auto prefetch_prev = [&](int t) {
    if (prev_y == nullptr) return;
    int tt = t;
    const int bx = tt % tiles_x; tt /= tiles_x;
    const int by = tt % tiles_y; tt /= tiles_y;
    const int bz = tt % tiles_z; tt /= tiles_z;
    const float* pybase = prev_y + (long)tt * nvox * 16;
    const float* pnbase = prev_norm ? prev_norm + (long)tt * nvox : nullptr;
    const int gx = bx * TX + li;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = wave * 4 + j;
      const int gz = bz * p3::TZ + (r >> 3), gy = by * p3::TY + (r & 7);
      const bool ok = gx < W && gy < H && gz < D;
      const int vox = ok ? (gz * H + gy) * W + gx : 0;
      pyv[j] = ok ? *(const f32x4*)(pybase + vox * 16 + cq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      pnv[j] = (ok && pnbase) ? pnbase[vox] : 1.f;
    }
  };

  auto compute = [&](int buf) {
    const float* base = lds + buf * p3::BUF_FLOATS;
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bcur[4], bnext[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bcur[j] = *(const f32x4*)(base + rowoffs[j]);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      if (tap + 1 < 27) {
        const int t1 = tap + 1;
        const int kx = t1 % 3, ky = (t1 / 3) % 3, kz = t1 / 9;
        const int toff = ((kz * p3::HY + ky) * p3::HX + kx) * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) bnext[j] = *(const f32x4*)(base + rowoffs[j] + toff);
      }
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][s4], bcur[j][s4], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) bcur[j] = bnext[j];
    }
  };

  // fused epilogue of tile t: He scale + bias, LeakyReLU, PixelNorm over the 16 channels, float4 store
  auto epilogue = [&](int t) {
    int tt = t;
    const int bx = tt % tiles_x; tt /= tiles_x;
    const int by = tt % tiles_y; tt /= tiles_y;
    const int bz = tt % tiles_z; tt /= tiles_z;
    const int n = tt;
    float* ybase = y + (long)n * nvox * 16;
    float* nbase = norm_out ? norm_out + (long)n * nvox : nullptr;
    const int gx = bx * TX + li;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = wave * 4 + j;
      const int gz = bz * p3::TZ + (r >> 3), gy = by * p3::TY + (r & 7);
      if (prev_y != nullptr) {
        // data-gradient launch: fold the previous layer's LeakyReLU' * PixelNorm' into the store
        const bool ok = gx < W && gy < H && gz < D;
        const int vox = ok ? (gz * H + gy) * W + gx : 0;
        const f32x4 yp = pyv[j];
        f32x4 g = acc[j] * he;
        if (prev_flags & LF_EPI_PIXELNORM) {
          float dot = g[0] * yp[0] + g[1] * yp[1] + g[2] * yp[2] + g[3] * yp[3];
          dot += __shfl_xor(dot, 16, 64);
          dot += __shfl_xor(dot, 32, 64);
          dot *= (1.f / 16.f);
          const float rinv = 1.0f / pnv[j];
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = (g[e] - yp[e] * dot) * rinv;
        }
        if (prev_flags & LF_EPI_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) g[e] = yp[e] > 0.f ? g[e] : g[e] * slope;
        }
        if (ok) *(f32x4*)(ybase + vox * 16 + cq * 4) = g;
        continue;
      }
      f32x4 v;
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = acc[j][e] * he + bv4[e];
        if (flags & LF_EPI_LRELU) u = fmaxf(u, u * slope);        // slope in (0,1): max(u, slope*u) == lrelu
        v[e] = u;
        ss += u * u;
      }
      float r_ = 1.f;
      if (flags & LF_EPI_PIXELNORM) {
        ss += __shfl_xor(ss, 16, 64);
        ss += __shfl_xor(ss, 32, 64);
        r_ = sqrtf(ss / 16.f + eps);
        const float rinv = 1.0f / r_;
        v[0] *= rinv; v[1] *= rinv; v[2] *= rinv; v[3] *= rinv;
      }
      if (gx < W && gy < H && gz < D) {
        const int vox = (gz * H + gy) * W + gx;
        *(f32x4*)(ybase + vox * 16 + cq * 4) = v;
        if ((flags & LF_EPI_PIXELNORM) && nbase != nullptr && cq == 0) nbase[vox] = r_;
      }
    }
  };

  // The two waves that share a SIMD (w and w+4) run their VALU-heavy epilogue at opposite ends of
  // the inter-barrier interval, so one wave's epilogue always overlaps its partner's MFMAs:
  //   waves 0-3:  [MFMA tile t][epilogue t]       | barrier
  //   waves 4-7:  [epilogue t-1][MFMA tile t]     | barrier
  int cur = 0;
  if (wave < 4) {
    for (int t = t_begin; t < t_end; ++t) {
      if (t + 1 < t_end) issue_dma(t + 1, cur ^ 1);               // lands during the MFMAs below
      prefetch_prev(t);
      compute(cur);
      epilogue(t);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // next tile's DMA has landed
      __syncthreads();
      cur ^= 1;
    }
  } else {
    for (int t = t_begin; t < t_end; ++t) {
      if (t + 1 < t_end) issue_dma(t + 1, cur ^ 1);
      if (t > t_begin) epilogue(t - 1);
      prefetch_prev(t);
      compute(cur);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      cur ^= 1;
    }
    epilogue(t_end - 1);
  }
}

// ---- pointwise convolution: GEMM over pixels, depth axis optionally folded into K ------------
template <int NT>
__global__ void __launch_bounds__(256) conv1x1_kernel(
    const float* __restrict__ x, const float* __restrict__ wpack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int P, int Cin, int ksl, long x_batch_stride, long x_slice_stride, int Cout, int Kp,
    long y_batch_stride, int y_row_stride, int y_slice_channels, long y_slice_stride,
    float he, unsigned flags, float slope, float eps,
    const float* __restrict__ prev_y, const float* __restrict__ prev_norm, unsigned prev_flags,
    float* __restrict__ amax_out, const float* __restrict__ xscale) {
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
  auto load_b = [&](int kc) -> f32x4 {
    const int k0 = kc * 16 + cq * 4;
    f32x4 bfrag = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (live && k0 < K) {
      const int s = k0 / Cin, c0 = k0 - s * Cin;      // ksl > 1 implies Cin % 4 == 0: no straddling
      const float* src = xrow + (long)s * x_slice_stride + c0;
      if (vec_in) {
        bfrag = *(const f32x4*)src;
        // (lf_conv1x1_fwd_scaled) one factor per (sample, slice, pixel): the product is rounded as the separate scaling pass
        // rounded it (lf_column_scale_fwd), so the sums are the same bits without the scaled volume ever existing
        if (xscale != nullptr) bfrag = bfrag * xscale[((long)n * ksl + s) * P + p];
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (k0 + e < K) bfrag[e] = src[e];
      }
    }
    return bfrag;
  };
  auto mac = [&](int kc, const f32x4& bfrag) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const f32x4 afrag = *(const f32x4*)(wrow + (long)t * 16 * Kp + kc * 16);
#pragma unroll
      for (int st = 0; st < 4; ++st)
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[st], bfrag[st], acc[t][0], 0, 0, 0);
    }
  };
  int kc = 0;
  if (NT == 1) {
    // long contractions (the factor projection: K = C * D = 2048) stream their activations from HBM with one load per
    // wave in flight; four chunks are requested before the first is consumed (same accumulation order: same bits;
    // measured 0.218 -> 0.209 ms per 8 x 128^3 x 16 launch -- occupancy already covered most of the latency)
    for (; kc + 4 <= nchunks; kc += 4) {
      f32x4 b4[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) b4[u] = load_b(kc + u);
#pragma unroll
      for (int u = 0; u < 4; ++u) mac(kc + u, b4[u]);
    }
  }
  for (; kc < nchunks; ++kc) mac(kc, load_b(kc));
  long rowidx[1] = {live ? ((long)n * P + p) : -1};
  long rowoff[1] = {live ? ((long)n * y_batch_stride + p * y_row_stride) : -1};
  const bool vec_out = ((y_row_stride | y_slice_channels) & 3) == 0 && ((y_batch_stride | y_slice_stride) & 3) == 0;
  epilogue_store<NT, 1>(acc, rowoff, rowidx, bias, y, norm_out, Cout, co_base, y_slice_channels, y_slice_stride,
                        vec_out, he, flags, slope, eps, prev_y, prev_norm, prev_flags, amax_out);
}

}  // namespace

extern "C" int lf_conv3x3_cout_padded(int Cout) {
  lf_clear_error();
  const int c16 = (Cout + 15) & ~15;
  if (c16 <= 16) return 16;
  if (c16 <= 32) return 32;
  return (c16 + 63) & ~63;
}

extern "C" int lf_conv1x1_cout_padded(int Cout) {
  lf_clear_error();
  const int c16 = (Cout + 15) & ~15;
  if (c16 <= 16) return 16;
  if (c16 <= 32) return 32;
  if (c16 <= 64) return 64;
  return (c16 + 127) & ~127;
}

static int conv3x3_launch(const float* x, const float* wpack, const float* bias, float* y, float* norm_out,
                          int dims, int N, int D, int H, int W, int Cin, int Cout,
                          float he, unsigned flags, float slope, float eps, void* stream,
                          const float* prev_y, const float* prev_norm, unsigned prev_flags) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return LF_EINVAL;
  if (prev_y != nullptr) {
    // fused previous-layer epilogue backward: plain data-gradient launch onto whole records of 16, 32 or 64 channels
    // (all channels of a voxel in one workgroup's tiles: the PixelNorm' dot product does not cross workgroups)
    if (flags != 0 || bias != nullptr || (Cout != 16 && Cout != 32 && Cout != 64) || !lf_aligned16(prev_y)) return LF_EINVAL;
    if ((prev_flags & LF_EPI_PIXELNORM) && prev_norm == nullptr) return LF_EINVAL;
  }
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
  if (Cin == 16 && Cout == 16 && dims == 3 && (long)D * H * W >= 4096 && (long)D * H * W * 64 < 0x7fffffffL &&
      slope > 0.f && slope < 1.f && (bias == nullptr || lf_aligned16(bias))) {
    // persistent double-buffered kernel: one 512-thread workgroup per CU
    const int ptx = (W + TX - 1) / TX, pty = (H + p3::TY - 1) / p3::TY, ptz = (D + p3::TZ - 1) / p3::TZ;
    const long pt = (long)ptx * pty * ptz * N;
    if (pt <= 0x7fffffffL) {
      static int cus = 0;                                         // CU count of the current device (cached)
      if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
          cus = v;
        else
          cus = 256;
      }
      const unsigned pgrid = (unsigned)(pt < cus ? pt : cus);
      hipLaunchKernelGGL(conv3d_c16_persistent_kernel, dim3(pgrid), dim3(p3::NTHREADS), 0, s, x, wpack, bias, y, norm_out,
                         N, D, H, W, ptx, pty, ptz, (int)pt, he, flags, slope, eps, prev_y, prev_norm, prev_flags);
      return lf_launch_status();
    }
  }
  if (CinP == 16 && CoutP == 16) {          // register-resident weights, unrolled + pipelined taps
    if (dims == 3)
      hipLaunchKernelGGL((conv3x3_c16_kernel<3>), grid, block, 0, s, x, wpack, bias, y, norm_out, N, D, H, W, Cin, Cout,
                         tiles_x, tiles_y, tiles_z, he, flags, slope, eps, prev_y, prev_norm, prev_flags);
    else
      hipLaunchKernelGGL((conv3x3_c16_kernel<2>), grid, block, 0, s, x, wpack, bias, y, norm_out, N, D, H, W, Cin, Cout,
                         tiles_x, tiles_y, tiles_z, he, flags, slope, eps, prev_y, prev_norm, prev_flags);
    return lf_launch_status();
  }
#define LAUNCH(DM, T) hipLaunchKernelGGL((conv3x3_kernel<DM, T>), grid, block, 0, s, x, wpack, bias, y, norm_out, \
                                         N, D, H, W, Cin, Cout, CinP, CoutP, tiles_x, tiles_y, tiles_z, he, flags, slope, eps, \
                                         prev_y, prev_norm, prev_flags)
  if (dims == 3) { if (NT == 4) LAUNCH(3, 4); else if (NT == 2) LAUNCH(3, 2); else LAUNCH(3, 1); }
  else           { if (NT == 4) LAUNCH(2, 4); else if (NT == 2) LAUNCH(2, 2); else LAUNCH(2, 1); }
#undef LAUNCH
  return lf_launch_status();
}

extern "C" int lf_conv3x3_fwd(const float* x, const float* wpack, const float* bias, float* y, float* norm_out,
                              int dims, int N, int D, int H, int W, int Cin, int Cout,
                              float he, unsigned flags, float slope, float eps, void* stream) {
  lf_clear_error();
  return conv3x3_launch(x, wpack, bias, y, norm_out, dims, N, D, H, W, Cin, Cout, he, flags, slope, eps, stream,
                        nullptr, nullptr, 0);
}

extern "C" int lf_conv3x3_bwd_data(const float* gy, const float* wpack_t, float* gx, int dims, int N, int D, int H, int W,
                                   int Cin, int Cout, float he, const float* prev_y, const float* prev_norm,
                                   unsigned prev_flags, float slope, void* stream) {
  lf_clear_error();
  return conv3x3_launch(gy, wpack_t, nullptr, gx, nullptr, dims, N, D, H, W, Cin, Cout, he, 0, slope, 0.f, stream,
                        prev_y, prev_norm, prev_flags);
}

static int conv1x1_launch(const float* x, const float* wpack, const float* bias, float* y, float* norm_out,
                          int N, int P, int Cin, int ksl, long x_batch_stride, long x_slice_stride,
                          int Cout, long y_batch_stride, int y_row_stride, int y_slice_channels,
                          long y_slice_stride,
                          float he, unsigned flags, float slope, float eps, void* stream,
                          const float* prev_y, const float* prev_norm, unsigned prev_flags, float* amax_out = nullptr,
                          const float* xscale = nullptr) {
  if (N <= 0 || P <= 0 || Cin <= 0 || ksl <= 0 || Cout <= 0 || y_slice_channels <= 0) return LF_EINVAL;
  if (xscale != nullptr && (Cin & 3)) return LF_EALIGN;
  if (prev_y != nullptr) {
    if (flags != 0 || bias != nullptr || y_slice_channels != 16 || y_row_stride != 16 || (Cout & 15) ||
        !lf_aligned16(prev_y) || ((y_batch_stride | y_slice_stride) & 15))
      return LF_EINVAL;
    if ((prev_flags & LF_EPI_PIXELNORM) && prev_norm == nullptr) return LF_EINVAL;
    if ((prev_flags & LF_EPI_DOT) && (prev_flags != LF_EPI_DOT || prev_norm == nullptr)) return LF_EINVAL;
  }
  if (y_row_stride < (y_slice_channels < Cout ? y_slice_channels : Cout)) return LF_EINVAL;
  if (ksl > 1 && (Cin & 3)) return LF_EALIGN;     // a lane's 4-channel group must not straddle slices
  if ((flags & LF_EPI_PIXELNORM) && Cout > 128) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(y) || !lf_aligned16(wpack)) return LF_EALIGN;
  if (y == nullptr && !(prev_y != nullptr && (prev_flags & LF_EPI_DOT))) return LF_EINVAL;
  if ((Cin & 3) == 0 && ((x_batch_stride | x_slice_stride) & 3)) return LF_EALIGN;
  const int K = ksl * Cin, Kp = (K + 15) & ~15;
  const int CoutP = lf_conv1x1_cout_padded(Cout);
  const int ntiles = CoutP / 16;
  // 1, 2, 4 or 8 output tiles per workgroup (CoutP is padded accordingly).  The unfolding data gradient with a fused
  // previous-layer backward (factor projection: one tile = one depth slice, 2.1 GB per launch) streams best at 4 slices per
  // workgroup: 0.536 / 0.511 / 0.510 / 0.582 ms at 8 / 4 / 2 / 1 (round 4, standalone at 8 x 128^3 x 16); requesting the saved
  // activation records AHEAD of the MFMA phase instead made it 0.82 ms (all waves of a CU then read, compute and write in
  // step: the staggered phases of the plain form keep reads and writes mixed at the HBM) -- not kept
  const int NT = ntiles >= 8 ? ((prev_y != nullptr && ntiles >= 16) ? 4 : 8) : ntiles;
  const int groups = ntiles / NT;
  dim3 grid((unsigned)((P + 63) / 64), groups, N), block(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T) hipLaunchKernelGGL((conv1x1_kernel<T>), grid, block, 0, s, x, wpack, bias, y, norm_out, \
                                     P, Cin, ksl, x_batch_stride, x_slice_stride, Cout, Kp, y_batch_stride, y_row_stride, \
                                     y_slice_channels, y_slice_stride, he, flags, slope, eps, prev_y, prev_norm, prev_flags, \
                                     amax_out, xscale)
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

extern "C" int lf_conv1x1_fwd(const float* x, const float* wpack, const float* bias, float* y, float* norm_out,
                              int N, int P, int Cin, int ksl, long x_batch_stride, long x_slice_stride,
                              int Cout, long y_batch_stride, int y_row_stride, int y_slice_channels,
                              long y_slice_stride,
                              float he, unsigned flags, float slope, float eps, void* stream) {
  lf_clear_error();
  return conv1x1_launch(x, wpack, bias, y, norm_out, N, P, Cin, ksl, x_batch_stride, x_slice_stride, Cout,
                        y_batch_stride, y_row_stride, y_slice_channels, y_slice_stride, he, flags, slope, eps, stream,
                        nullptr, nullptr, 0);
}

extern "C" int lf_conv1x1_fwd_scaled(const float* x, const float* xscale, const float* wpack, const float* bias, float* y,
                                     float* norm_out, int N, int P, int Cin, int ksl, long x_batch_stride, long x_slice_stride,
                                     int Cout, long y_batch_stride, int y_row_stride, int y_slice_channels, long y_slice_stride,
                                     float he, unsigned flags, float slope, float eps, void* stream) {
  lf_clear_error();
  if (xscale == nullptr) return LF_EINVAL;
  return conv1x1_launch(x, wpack, bias, y, norm_out, N, P, Cin, ksl, x_batch_stride, x_slice_stride, Cout,
                        y_batch_stride, y_row_stride, y_slice_channels, y_slice_stride, he, flags, slope, eps, stream,
                        nullptr, nullptr, 0, nullptr, xscale);
}

extern "C" int lf_conv1x1_bwd_data(const float* gy, const float* wpack_t, float* gx, int N, int P, int Cin, int Cout,
                                   long y_batch_stride, int y_row_stride, int y_slice_channels, long y_slice_stride,
                                   float he, const float* prev_y, const float* prev_norm, unsigned prev_flags,
                                   float slope, float* amax_out, void* stream) {
  lf_clear_error();
  return conv1x1_launch(gy, wpack_t, nullptr, gx, nullptr, N, P, Cin, 1, (long)P * Cin, 0, Cout, y_batch_stride,
                        y_row_stride, y_slice_channels, y_slice_stride, he, 0, slope, 0.f, stream, prev_y, prev_norm,
                        prev_flags, amax_out);
}
