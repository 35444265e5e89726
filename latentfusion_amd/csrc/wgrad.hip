// Weight (and bias) gradients of the He-equalised convolutions -- the training-step side of
//   Equalized.forward   latentfusion/modules/equalized.py:57-64   (y = conv(x, W) * he + b)
//   Block.forward       latentfusion/modules/blocks.py:152-158
// which the reference gets from autograd (tools/train/train_reconstruct.py:421-535).
//
//   gw[tap][co][ci] = scale * sum_v gpre[v][co] * x[v + tap][ci]         (zero padding, channels-last)
//
// i.e. one [Cout x V] x [V x Cin] product per tap with the voxel axis as the contraction, on
// v_mfma_f32_16x16x4_f32: A = gpre^T (lane (m, k) reads channel m of voxel k), B = shifted x.  Every block
// reduces a contiguous run of voxels for one (tap, 16x16 channel tile); block partials are summed in a
// fixed order in fp64 by a second kernel, so the result does not depend on scheduling.
// x == NULL stands for an all-ones single-channel input: gw[0][co][0] is then the bias gradient.
#include <type_traits>
#include "lf_common.h"
#include "ring_tile.h"

namespace {

template <int DIMS>
__global__ void __launch_bounds__(256) wgrad_partial_kernel(
    const float* __restrict__ x, const float* __restrict__ gp, float* __restrict__ partial,
    int N, int D, int H, int W, int Cin, int Cout, int chunk, int ncit) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m = lane & 15, k = lane >> 4;
  const int tap = blockIdx.y;
  const int ct = blockIdx.z / ncit, cit = blockIdx.z - ct * ncit;
  int dz = 0, dy = 0, dx = 0;
  if (DIMS == 3) { dz = tap / 9 - 1; dy = (tap / 3) % 3 - 1; dx = tap % 3 - 1; }
  if (DIMS == 2) { dy = tap / 3 - 1; dx = tap % 3 - 1; }
  const int total = N * D * H * W;
  const int v_begin = blockIdx.x * chunk;
  const int v_end = min(v_begin + chunk, total);
  const int co = ct * 16 + m, ci = cit * 16 + m;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int v0 = v_begin + wave * 4; v0 < v_end; v0 += 16) {
    const int v = v0 + k;
    float a = 0.f, b = 0.f;
    if (v < v_end) {
      if (co < Cout) a = gp[(long)v * Cout + co];
      if (x == nullptr) {
        b = (m == 0) ? 1.f : 0.f;
      } else if (ci < Cin) {
        const int r1 = v / W, px = v - r1 * W;
        const int r2 = r1 / H, py = r1 - r2 * H;
        const int n = r2 / D, pz = r2 - n * D;
        const int sx = px + dx, sy = py + dy, sz = pz + dz;
        if ((unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H && (unsigned)sz < (unsigned)D)
          b = x[(long)(((n * D + sz) * H + sy) * W + sx) * Cin + ci];
      }
    }
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  __shared__ float red[4][256];
#pragma unroll
  for (int i = 0; i < 4; ++i) red[wave][(4 * k + i) * 16 + m] = acc[i];        // D[row 4k+i][col m]
  __syncthreads();
  const int t = threadIdx.x;
  const float s = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
  partial[(((long)blockIdx.x * gridDim.y + tap) * gridDim.z + blockIdx.z) * 256 + t] = s;
}

// gw[tap][co][ci] = scale * sum over blocks (fixed order, fp64).  grid (taps, channel tiles, 4): a workgroup sums 64 of a
// tile's 256 outputs, four interleaved runs of blocks in parallel (one per wave, eight loads in flight each), combined
// as (s0 + s1) + (s2 + s3): one serial walk per output took 60-120 us for 256-512 blocks, latency-bound.
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                           int nblk, int taps, int ntiles, int ncit, int Cin, int Cout,
                                                           float scale) {
  const int tap = blockIdx.x, tile = blockIdx.y, t = blockIdx.z * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
  const long stride = (long)taps * ntiles * 256;
  const float* src = partial + ((long)tap * ntiles + tile) * 256 + t;
  double s = 0.0;
  int b = g;
  for (; b + 28 < nblk; b += 32) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = src[(long)(b + 4 * u) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (double)v[u];
  }
  for (; b < nblk; b += 4) s += (double)src[(long)b * stride];
  __shared__ double red[4][64];
  red[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0) {
    const int o = threadIdx.x;
    const double tot = (red[0][o] + red[1][o]) + (red[2][o] + red[3][o]);
    const int ct = tile / ncit, cit = tile - ct * ncit;
    const int co = ct * 16 + (t >> 4), ci = cit * 16 + (t & 15);
    if (co < Cout && ci < Cin) gw[((long)tap * Cout + co) * Cin + ci] = (float)(tot * (double)scale);
  }
}


// ---- 3-D, 16 -> 16 channels: persistent LDS-staged kernel -------------------------------------------
// One 512-thread workgroup per CU walks 2 x 8 x 16-voxel tiles: the x halo (4 x 10 x 18 voxels) and the gpre
// tile arrive by LDS-DMA (double-buffered, zero fill outside the volume), every wave owns two x-rows of the
// tile and keeps all 27 tap accumulators (108 VGPRs) across its whole tile range; the per-wave partial sums
// are combined through LDS at the end, one block per workgroup is written and wgrad_reduce_kernel sums the blocks
// in a fixed order.
constexpr int WTZ = 2, WTY = 8, WTX = 16, WHZ = 4, WHY = 10, WHX = 18;
constexpr int WHALO = WHZ * WHY * WHX;                          // 720 voxels
constexpr int WPH = (WHALO * 4 + 63) / 64;                      // 45 halo pieces (1 KiB)
constexpr int WPG = WTZ * WTY * WTX * 4 / 64;                   // 16 gpre pieces
constexpr int WBUF = (WPH + WPG) * 1024;                        // 62,464 B per buffer
constexpr int WNIT = (WPH + WPG + 7) / 8;                       // 8 pieces per wave

__global__ void __launch_bounds__(512) wgrad3d_c16_kernel(
    const float* __restrict__ x, const float* __restrict__ gp, float* __restrict__ partial,
    int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, k = lane >> 4;
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;   // XCD-aware ranges
  const int per = (ntiles + nb - 1) / nb;
  const int t_begin = lb * per;
  const int t_end = min(t_begin + per, ntiles);
  const long nvox = (long)D * H * W;
  const unsigned sample_bytes = (unsigned)(nvox * 64);

  f32x4 acc[27];
#pragma unroll
  for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (t_begin < t_end) {
    // DMA piece constants: local voxel coordinates + 16-byte quarter; halo pieces first, then gpre pieces
    int lxyzq[WNIT];
#pragma unroll
    for (int it = 0; it < WNIT; ++it) {
      const int s = wave + 8 * it;
      const int p = (s < WPH ? s : s - WPH) * 64 + lane;
      const int v = p >> 2, q = p & 3;
      int lx, ly, lz;
      bool ok;
      if (s < WPH) { lx = v % WHX; ly = (v / WHX) % WHY; lz = v / (WHX * WHY); ok = v < WHALO; }
      else         { lx = v & 15; ly = (v >> 4) & 7; lz = v >> 7; ok = s < WPH + WPG; }
      lxyzq[it] = ok ? (lx | (ly << 8) | (lz << 16) | (q << 24)) : -1;
    }
    auto issue = [&](int t, int bufsel) {
      int tt = t;
      const int bx = tt % tiles_x; tt /= tiles_x;
      const int by = tt % tiles_y; tt /= tiles_y;
      const int bz = tt % tiles_z; tt /= tiles_z;
      __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)tt * nvox * 16), 0, sample_bytes, 0x00020000);
      __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(gp + (long)tt * nvox * 16), 0, sample_bytes, 0x00020000);
      unsigned char* dst = smem + bufsel * WBUF;
#pragma unroll
      for (int it = 0; it < WNIT; ++it) {
        const int s = wave + 8 * it;
        if (s >= WPH + WPG) continue;                              // wave-uniform
        const int halo = s < WPH ? 1 : 0;
        const int gx = bx * WTX - halo + (lxyzq[it] & 0xff), gy = by * WTY - halo + ((lxyzq[it] >> 8) & 0xff),
                  gz = bz * WTZ - halo + ((lxyzq[it] >> 16) & 0xff);
        const bool ok = (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D && lxyzq[it] >= 0;
        const int voff = ok ? ((gz * H + gy) * W + gx) * 64 + ((lxyzq[it] >> 24) << 4) : 0x7fffffff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(halo ? rx : rg, (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, voff, 0, 0, 0);
      }
    };
    // per-lane LDS byte offsets: A = gpre[voxel k of the group][channel m], B = x[same voxel + tap][channel m]
    const int a_lane = WPH * 1024 + ((2 * wave) * 16 + k) * 64 + m * 4;                 // + (j*16 + 4*xg)*64
    const int rz = (2 * wave) >> 3, ry = (2 * wave) & 7;                               // first of the wave's two rows
    const int b_lane = ((rz * WHY + ry) * WHX + k) * 64 + m * 4;                       // + (j*18 + 4*xg + tap offset)*64

    issue(t_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
      const int cur = (t - t_begin) & 1;
      const unsigned char* buf = smem + cur * WBUF;
      if (t + 1 < t_end) issue(t + 1, cur ^ 1);
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const int j = g >> 2, xg = g & 3;                             // row j of the wave's pair, 4-voxel group xg
        const float a = *(const float*)(buf + a_lane + (j * 16 + 4 * xg) * 64);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
          const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
          const float b = *(const float*)(buf + b_lane + (((kz * WHY) + ky + j) * WHX + 4 * xg + kx) * 64);
          acc[tap] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[tap], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  // the eight waves' partials are summed through LDS in a fixed tree (waves w+4 -> w, w+2 -> w, 1 -> 0), so a
  // workgroup writes ONE 27 x 256 block (wgrad_reduce_kernel then sums `gridDim.x` blocks instead of 8x as many:
  // 7 MB instead of 57 MB per launch, which dominated small launches)
  f32x4* red = (f32x4*)smem;                                    // [wave slot][27][64 lanes]
#pragma unroll
  for (int half = 4; half >= 1; half >>= 1) {
    __syncthreads();                                            // previous round consumed / tile loop done with LDS
    if (wave >= half && wave < 2 * half) {
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) red[((wave - half) * 27 + tap) * 64 + lane] = acc[tap];
    }
    __syncthreads();
    if (wave < half) {
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) acc[tap] += red[(wave * 27 + tap) * 64 + lane];
    }
  }
  if (wave == 0) {
    float* out = partial + (long)blockIdx.x * 27 * 256;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[tap * 256 + (4 * k + i) * 16 + m] = acc[tap][i];
  }
}

// ---- the same product on the bf16 MFMA (autocast policy of the training step) ------------------------------------------
// v_mfma_f32_16x16x32_bf16 contracts 32 voxels per instruction at 8x the fp32 MFMA's rate per voxel, but wants its K
// run -- 8 consecutive voxels of ONE channel per lane -- where channels-last memory has 16 channels of one voxel: the
// tile is transposed on its way into LDS.  A thread loads the same 16-byte quarter (4 channels) of TWO voxels adjacent
// in x into registers, rounds to bf16 (RNE, v_cvt_pk_bf16_f32: the identity on operands autocast has already rounded),
// and writes four dwords = the two voxels of each of its channels (v_perm) into channel-major planes.  A lane then reads
// its 8-voxel K run with one ds_read_b128; the kx = 0, 1, 2 taps of a row come from the SAME five dwords (kx = 2: a dword
// later; kx = 1: v_alignbyte).
// With the MFMAs this cheap the kernel is bound by what it asks of L2 / HBM, so a workgroup walks UP a column of
// 2 x 8 x 16 tiles and keeps the x halo in a ring of six z-plane slots ([16 ch][6 slots][10 rows][18 (+6) x] bf16): a
// tile reads four planes, two of which the next tile reads again, so only the two new planes (and the tile's own 2 x 8 x 16
// block of gpre, double-buffered) are fetched per tile -- 39 KB instead of 62 KB -- while the current tile is contracted.
// Four waves; wave w owns the (kz, ky) stencil rows 2w, 2w+1 for all eight 32-voxel K groups (2 rows x 16 voxels) of the
// tile and row 8 for K groups 2w, 2w+1: 54 MFMAs per wave and tile, nine accumulators, and only row 8 is summed across
// waves (through LDS, fixed order) at the end.  Two workgroups per CU (62 KB of LDS each).
// Products of bf16 operands are exact in fp32 and accumulation is fp32 as on the fp32 kernel: same result up to
// summation order.  Partials: one 27 x 256 block per workgroup, summed by wgrad_reduce_kernel (fixed order, fp64).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int I> struct WIC { static constexpr int v = I; };
template <int B, int E, typename F>
__device__ __forceinline__ void wstatic_for(F&& f) {
  if constexpr (B < E) { f(WIC<B>{}); wstatic_for<B + 1, E>(f); }
}
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#ifndef WG_ABL
#define WG_ABL 0                                                 // ablations (tools/wgrad_ab.py): 1 no contraction, 2 no commit to LDS, 4 no loads
#endif
constexpr int BRING = 6;                                         // z-plane slots: 4 being read + 2 being filled
constexpr int BRS = 48;                                          // bytes per halo row: 18 bf16 (+6)
constexpr int BPL = WHY * BRS;                                   // 480 B per plane slot and channel
constexpr int BXS = BRING * BPL + 16;                            // channel stride of the x planes (2896 B: 20 banks)
constexpr int BGR = 32;                                          // bytes per gpre row: 16 bf16
constexpr int BGS = WTZ * WTY * BGR + 16;                        // channel stride of the gpre planes (528 B)
constexpr int BXB = 16 * BXS;                                    // 46,336 B
constexpr int BGB = 16 * BGS;                                    // 8,448 B per gpre buffer
constexpr int BLDS = BXB + 2 * BGB;                              // 63,232 B
constexpr int BPP = 2 * WHY * (WHX / 2);                         // 180 voxel pairs in two halo planes
constexpr int BXIT = (BPP * 4 + 255) / 256;                      // 3 x pair-pieces per thread and tile ...
constexpr int BNIT = BXIT + (WTZ * WTY * WTX / 2) * 4 / 256;     // ... + 2 gpre pair-pieces
constexpr unsigned BOOB = 0x80000000u;

__device__ __forceinline__ int bmod6(int v) { return v >= 6 ? v - 6 : v; }      // v in [0, 12)

// IO (round 5): bit 0 -- x, bit 1 -- gpre are stored as bf16 channels-last records (32 B per voxel) instead of fp32 ones: the
// staging rounds fp32 operands to bf16 anyway, so operands kept rounded in memory give the same products from half the bytes.
template <int IO>
__global__ void __launch_bounds__(256, 2) wgrad3d_c16_bf16_kernel(
    const float* __restrict__ x, const float* __restrict__ gp, float* __restrict__ partial,
    int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z, int ntiles) {
  constexpr bool X16 = (IO & 1) != 0, G16 = (IO & 2) != 0;
  constexpr int XSH = X16 ? 5 : 6, GSH = G16 ? 5 : 6;                // log2 bytes per voxel record
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, k = lane >> 4;
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;   // XCD-aware ranges
  const int per = (ntiles + nb - 1) / nb;
  const int t_begin = lb * per;
  const int t_end = min(t_begin + per, ntiles);
  const long nvox = (long)D * H * W;
  const unsigned plane_vox = (unsigned)(H * W);

  f32x4 acc[9];                                                  // rows 2w, 2w+1 (x 3 kx), then this wave's share of row 8
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (t_begin < t_end) {
    const int q = lane & 3;
    // pair-piece constants.  x pieces (it < 3): pair P of the two incoming planes = (plane P / 90, row, x pair); gpre pieces:
    // pair of the tile's 2 x 8 x 16 block.  lxy = x | y << 8 | plane << 16 relative to the tile origin - 1 (x) / the tile
    // origin (gpre); dst = LDS byte offset of channel 4q's dword inside slot 0 / gpre buffer 0.  The slots past the 180th
    // pair repeat it (same data, same address), so nothing below branches on the wave.
    int lxy[BNIT], dst[BNIT];
#pragma unroll
    for (int it = 0; it < BNIT; ++it) {
      if (it < BXIT) {
        const int pr = min((wave + 4 * it) * 16 + (lane >> 2), BPP - 1);
        const int pl = pr / (BPP / 2), rem = pr - pl * (BPP / 2), row = rem / (WHX / 2), px = rem - row * (WHX / 2);
        lxy[it] = (2 * px) | (row << 8) | (pl << 16);
        dst[it] = 4 * q * BXS + row * BRS + px * 4;
      } else {
        const int pr = (wave + 4 * (it - BXIT)) * 16 + (lane >> 2);
        const int lx = 2 * (pr & 7), ly = (pr >> 3) & 7, lz = pr >> 6;
        lxy[it] = lx | (ly << 8) | (lz << 16);
        dst[it] = BXB + 4 * q * BGS + (lz * WTY + ly) * BGR + lx * 2;
      }
    }
    // per COLUMN of tiles: byte offsets inside the sample of each piece's first voxel with the incoming plane pair / the
    // tile at z = 0, and which of its two voxels lie inside the volume in x and y (bits 2 it, 2 it + 1); the plane itself is
    // added per tile, and planes outside [0, D) fall outside the buffer (below zero wraps) and read as zero
    unsigned coff[BNIT];
    unsigned okm = 0;
    const float *cx_ptr = x, *cg_ptr = gp;
    auto column = [&](int bx, int by, int bn) {
      okm = 0;
#pragma unroll
      for (int it = 0; it < BNIT; ++it) {
        const int halo = it < BXIT ? 1 : 0;
        const int gx = bx * WTX - halo + (lxy[it] & 0xff), gy = by * WTY - halo + ((lxy[it] >> 8) & 0xff);
        const int sh = halo ? XSH : GSH;
        coff[it] = (unsigned)(((gy * W + gx) << sh) + (q << (sh - 2))) + (((unsigned)(lxy[it] >> 16) * plane_vox) << sh);
        const bool oky = (unsigned)gy < (unsigned)H;
        okm |= ((oky && (unsigned)gx < (unsigned)W) ? 1u : 0u) << (2 * it);
        okm |= ((oky && (unsigned)(gx + 1) < (unsigned)W) ? 1u : 0u) << (2 * it + 1);
      }
      cx_ptr = x + (long)bn * nvox * (X16 ? 8 : 16);               // (declared float*: a bf16 record is 8 floats' worth of bytes)
      cg_ptr = gp + (long)bn * nvox * (G16 ? 8 : 16);
    };
    u32x4 st[BNIT][2];
    // x planes zx, zx + 1 of the column (and, with_g, the gpre block of the tile at planes zg, zg + 1) -> registers
    auto issue = [&](int zx, int zg, bool with_g) {
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)cx_ptr, 0, (unsigned)(nvox << XSH), 0x00020000);
      const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)cg_ptr, 0, with_g ? (unsigned)(nvox << GSH) : 0u, 0x00020000);
      const unsigned zoffx = ((unsigned)zx * plane_vox) << XSH, zoffg = ((unsigned)zg * plane_vox) << GSH;
#pragma unroll
      for (int it = 0; it < BNIT; ++it) {
        const unsigned o = coff[it] + (it < BXIT ? zoffx : zoffg);
        const int o0 = (int)(((okm >> (2 * it)) & 1u) ? o : BOOB);
        if ((it < BXIT) ? X16 : G16) {                               // (folds after unrolling)
          const int o1 = (int)(((okm >> (2 * it + 1)) & 1u) ? o + 32u : BOOB);
          const u32x2 a = __builtin_amdgcn_raw_buffer_load_b64(it < BXIT ? rx : rg, o0, 0, 0);
          const u32x2 b = __builtin_amdgcn_raw_buffer_load_b64(it < BXIT ? rx : rg, o1, 0, 0);
          st[it][0] = (u32x4){a[0], a[1], 0u, 0u};
          st[it][1] = (u32x4){b[0], b[1], 0u, 0u};
        } else {
          const int o1 = (int)(((okm >> (2 * it + 1)) & 1u) ? o + 64u : BOOB);
          st[it][0] = __builtin_amdgcn_raw_buffer_load_b128(it < BXIT ? rx : rg, o0, 0, 0);
          st[it][1] = __builtin_amdgcn_raw_buffer_load_b128(it < BXIT ? rx : rg, o1, 0, 0);
        }
      }
    };
    // registers -> bf16 planes: x pieces into ring slots s0 (first plane) / s1, gpre pieces into buffer gsel
    auto commit = [&](int s0, int s1, int gsel, bool with_g) {
#pragma unroll
      for (int it = 0; it < BNIT; ++it) {
        if (it >= BXIT && !with_g) continue;                       // (uniform)
        unsigned e01, e23, o01, o23;
        if ((it < BXIT) ? X16 : G16) {                               // already bf16 pairs
          e01 = st[it][0][0]; e23 = st[it][0][1]; o01 = st[it][1][0]; o23 = st[it][1][1];
        } else {
          const f32x4 e = __builtin_bit_cast(f32x4, st[it][0]), o = __builtin_bit_cast(f32x4, st[it][1]);
          e01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e[0], e[1]}, bf16x2));
          e23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){e[2], e[3]}, bf16x2));
          o01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){o[0], o[1]}, bf16x2));
          o23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){o[2], o[3]}, bf16x2));
        }
        const int cs = it < BXIT ? BXS : BGS;
        unsigned char* d = smem + dst[it] + (it < BXIT ? ((lxy[it] >> 16) ? s1 : s0) * BPL : gsel * BGB);
        // one dword per channel: [voxel x, voxel x + 1]
        *(unsigned*)(d) = __builtin_amdgcn_perm(o01, e01, 0x05040100u);
        *(unsigned*)(d + cs) = __builtin_amdgcn_perm(o01, e01, 0x07060302u);
        *(unsigned*)(d + 2 * cs) = __builtin_amdgcn_perm(o23, e23, 0x05040100u);
        *(unsigned*)(d + 3 * cs) = __builtin_amdgcn_perm(o23, e23, 0x07060302u);
      }
    };
    // operand offsets: K group kg = (plane z = kg >> 2, row pair rp = kg & 3); lane group k -> row 2*rp + (k >> 1), x half k & 1
    const int a_lane = BXB + m * BGS + (k >> 1) * BGR + (k & 1) * 16;                             // + gsel*BGB + (z*WTY + 2*rp) * BGR
    const int b_lane = m * BXS + (k >> 1) * BRS + (k & 1) * 16;                                   // + slot(z + kz)*BPL + (ky + 2*rp) * BRS
    const int kz0 = (2 * wave) / 3, ky0 = (2 * wave) % 3, kz1 = (2 * wave + 1) / 3, ky1 = (2 * wave + 1) % 3;
    auto three_taps = [&](const bf16x8 a, const unsigned char* row, f32x4* c) {
      const u32x4 d = *(const u32x4*)row;
      const unsigned d4 = *(const unsigned*)(row + 16);
      const u32x4 b1 = (u32x4){__builtin_amdgcn_alignbyte(d[1], d[0], 2), __builtin_amdgcn_alignbyte(d[2], d[1], 2),
                               __builtin_amdgcn_alignbyte(d[3], d[2], 2), __builtin_amdgcn_alignbyte(d4, d[3], 2)};
      const u32x4 b2 = (u32x4){d[1], d[2], d[3], d4};
      c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, d), c[0], 0, 0, 0);
      c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b1), c[1], 0, 0, 0);
      c[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, b2), c[2], 0, 0, 0);
    };
    auto contract = [&](int rot, int gsel) {
      // halo plane hp (0..3) of the tile sits in ring slot (rot + hp) % 6
      const unsigned char* ab = smem + a_lane + gsel * BGB;
      const unsigned char* r0[2], *r1[2], *r8[2];
#pragma unroll
      for (int z = 0; z < 2; ++z) {
        r0[z] = smem + b_lane + bmod6(rot + z + kz0) * BPL + ky0 * BRS;
        r1[z] = smem + b_lane + bmod6(rot + z + kz1) * BPL + ky1 * BRS;
        r8[z] = smem + b_lane + bmod6(rot + z + 2) * BPL + 2 * BRS;
      }
#pragma unroll
      for (int kg = 0; kg < 8; ++kg) {
        const int z = kg >> 2, rp = kg & 3;
        const bf16x8 a = *(const bf16x8*)(ab + (z * WTY + 2 * rp) * BGR);
        three_taps(a, r0[z] + 2 * rp * BRS, acc);
        three_taps(a, r1[z] + 2 * rp * BRS, acc + 3);
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {                               // stencil row 8 = (kz, ky) = (2, 2): K groups 2w, 2w + 1
        const int kg = 2 * wave + g, z = kg >> 2, rp = kg & 3;    // (wave-uniform)
        const bf16x8 a = *(const bf16x8*)(ab + (z * WTY + 2 * rp) * BGR);
        three_taps(a, (z ? r8[1] : r8[0]) + 2 * rp * BRS, acc + 6);
      }
    };

    // tile coordinates are stepped, not divided; z fastest: the range walks up columns
    int cx, cy, cz, cn;
    {
      int tt = t_begin;
      cz = tt % tiles_z; tt /= tiles_z;
      cx = tt % tiles_x; tt /= tiles_x;
      cy = tt % tiles_y; cn = tt / tiles_y;
    }
    int rot = 0;
    column(cx, cy, cn);
    issue(cz * WTZ - 1, 0, false);
    __builtin_amdgcn_sched_barrier(0);
    commit(0, 1, 0, false);
    issue(cz * WTZ + 1, cz * WTZ, true);
    __builtin_amdgcn_sched_barrier(0);
    commit(2, 3, 0, true);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
      const int gsel = (t - t_begin) & 1;
      int nx = cx, ny = cy, nz = cz + 1, nn = cn;
      if (nz == tiles_z) { nz = 0; ++nx; }
      if (nx == tiles_x) { nx = 0; ++ny; }
      if (ny == tiles_y) { ny = 0; ++nn; }
      const bool on = t + 1 < t_end;
      const bool slide = on && nz != 0;
      if (on) {
        if (!slide) column(nx, ny, nn);
        // the two planes the next tile adds (slide: its halo planes 2, 3; new column: its planes 0, 1) and its gpre block
        if (!(WG_ABL & 4)) issue(nz * WTZ - 1 + (slide ? 2 : 0), nz * WTZ, true);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!(WG_ABL & 1)) contract(rot, gsel);
      __builtin_amdgcn_sched_barrier(0);
      if (on && !(WG_ABL & 2)) commit(bmod6(rot + 4), bmod6(rot + 5), gsel ^ 1, true);
      __syncthreads();                                          // this tile's planes 0, 1 are free; the new planes are visible
      if (on && !slide) {
        // bottom of a new column: what arrived are its planes 0, 1 (slots rot+4, rot+5); planes 2, 3 go to the slots this
        // tile has just released (exposed once per column)
        rot = bmod6(rot + 4);
        issue(nz * WTZ + 1, 0, false);
        __builtin_amdgcn_sched_barrier(0);
        commit(bmod6(rot + 2), bmod6(rot + 3), 0, false);
        __syncthreads();
      } else {
        rot = bmod6(rot + 2);
      }
      cx = nx; cy = ny; cz = nz; cn = nn;
    }
  }
  // row 8 (taps 24..26): waves 1..3 hand their shares to wave 0 through LDS, summed in wave order
  f32x4* red = (f32x4*)smem;                                    // [wave - 1][3][64 lanes]
  __syncthreads();
  if (wave > 0) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) red[((wave - 1) * 3 + kx) * 64 + lane] = acc[6 + kx];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[6 + kx] += red[(w * 3 + kx) * 64 + lane];
  }
  float* out = partial + (long)blockIdx.x * 27 * 256;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r == 2 && wave != 0) break;
    const int zy = r == 2 ? 8 : 2 * wave + r;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(zy * 3 + kx) * 256 + (4 * k + i) * 16 + m] = acc[r * 3 + kx][i];
  }
}

// ---- round 6: the same contraction with the operands read by the LDS TRANSPOSE load (ds_read_b64_tr_b16) ------------------
// PMC on the kernel above: SQ_LDS_BANK_CONFLICT = 0.68 of SQ_LDS_IDX_ACTIVE, LDS issue stalls + waits = 0.7 of the wave cycles
// -- its channel-major planes (strides chosen for 32 banks; ds_read_b128 sees 64) conflict two-fold on every operand read,
// four-fold on the fifth-dword reads, and the transposing commit (v_perm + four ds_write_b32 per pair-piece) conflicts too;
// the contraction alone takes 1.67 of the 1.92 ms per 32 volumes (tools/wgrad_ab.py, -DWG_ABL).  gfx950 can transpose on the
// way OUT of LDS instead: the tile stays in LDS as it is in memory, bf16 channels-last records of 32 B (x: the six-slot
// ring of 10 x 18-voxel halo planes of the ring convolutions, conv_split.hip; gpre: the 2 x 8 x 16 block, double-buffered),
// written with plain 8-byte stores, and a lane gets "channel (lane & 15) of four consecutive voxels" from one
// ds_read_b64_tr_b16 at the address of record (voxel + (lane >> 2 & 3)), channel quad (lane & 3).  Two of them = the
// 8-voxel K run of v_mfma_f32_16x16x32_bf16; a stencil shift along x is 32 B of address, not a v_alignbyte.  Bank-conflict
// free by construction: an instruction's two 16-lane groups of a half-wave read the 128-byte chunks x = 0..3 and 4..7
// (then 8..11 and 12..15) of one row = disjoint halves of the 64 banks, whatever the row / tap offset -- so the K run of
// lane group k is voxels {4 (k & 1) .. +3} and {4 (k & 1) + 8 .. +3} of row k >> 1 of the K group's two rows, for BOTH
// operands.  Wave / accumulator assignment, partials and the reduce kernel are those of the kernel above (bit-compatible
// up to the order of the products inside one MFMA's K = 32, which is exact in fp32: same results).
typedef __bf16 bf16x4t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4t lds_bf16x4t;
constexpr int TXP = RINGs * PLANE_B;                              // 34,560 B: the x ring
constexpr int TGB = WTZ * WTY * WTX * 32;                         // 8,192 B per gpre buffer
constexpr int TLDS = TXP + 2 * TGB + 1024;                        // + guard: a shifted K run of the last row reads past its plane

__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p0, const unsigned char* p1) {
  const bf16x4t lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4t*)p0);
  const bf16x4t hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4t*)p1);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

#ifndef WG_ABL
#define WG_ABL 0                                                 // ablations of the kernel below (tools/wgrad_ab.py): 1 no contraction, 2 no commit to LDS, 4 no loads
#endif
#ifndef WG_TR_PF
#define WG_TR_PF 1                                                // contraction steps whose operand reads are in flight ahead of the MFMAs
#endif
#ifndef WG_TR_WGS
#define WG_TR_WGS 2                                               // resident workgroups per CU (8 waves each: 53 KB of LDS, <= 128 VGPRs)
#endif
template <int IO>
__global__ void __launch_bounds__(512, 2 * WG_TR_WGS) wgrad3d_c16_tr_kernel(
    const float* __restrict__ x, const float* __restrict__ gp, float* __restrict__ partial,
    int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z, int ntiles) {
  constexpr bool X16 = (IO & 1) != 0, G16 = (IO & 2) != 0;
  constexpr int XSH = X16 ? 5 : 6, GSH = G16 ? 5 : 6;
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, k = lane >> 4;
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
  const int per = (ntiles + nb - 1) / nb;
  const int t_begin = lb * per;
  const int t_end = min(t_begin + per, ntiles);
  const long nvox = (long)D * H * W;

  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (t_begin < t_end) {
    for (int i = tid; i < TLDS / 16; i += 512) ((u32x4*)smem)[i] = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();
    // ---- staging (waves 4..7).  x: staging wave w fetches rows 5 (w >> 1) .. +4 of incoming plane w & 1 as six pieces
    // (conv_split.hip); gpre: the tile's 256 voxels x 4 quarters = 4 pieces per staging thread, piece j of thread t = quarter
    // t & 3 of voxel (t >> 2) + 64 j
    const bool producer = wave >= 4;                                // (wave-uniform)
    const int pw = wave & 3, ptid = tid & 255;
    const int pzw = pw & 1, ryw = pw >> 1, frow0 = 5 * ryw;
    const int erow = frow0 + (lane >> 3), ecol = 16 + ((lane >> 2) & 1);
    const bool e_ok = lane < 40;
    const int eldso = e_ok ? (erow * HXs + ecol) * 32 + (lane & 3) * 8 : -1;
    const int xplane_b = (H * W) << XSH, gplane_b = (H * W) << GSH;
    int foff[NPIECE], goff[4];
    const unsigned char *fx = (const unsigned char*)x, *fg = (const unsigned char*)gp;
    auto column = [&](int bx, int by, int bn) {
      const int ox = bx * WTX - 1, oy = by * WTY - 1;
      const int col = ox + (lane >> 2);
#pragma unroll
      for (int it = 0; it < 5; ++it) {
        const int row = oy + frow0 + it;
        foff[it] = ((unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H) ? ((row * W + col) << XSH) + ((lane & 3) << (XSH - 2)) : OOB;
      }
      const int row = oy + erow, c2 = ox + ecol;
      foff[5] = (e_ok && (unsigned)c2 < (unsigned)W && (unsigned)row < (unsigned)H) ? ((row * W + c2) << XSH) + ((lane & 3) << (XSH - 2)) : OOB;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int v = (ptid >> 2) + 64 * j, lx = v & 15, ly = (v >> 4) & 7;      // (lz = v >> 7 goes into the plane offset)
        const int gx = bx * WTX + lx, gy = by * WTY + ly;
        goff[j] = (gx < W && gy < H) ? ((gy * W + gx) << GSH) + ((ptid & 3) << (GSH - 2)) : OOB;
      }
      fx = (const unsigned char*)x + ((long)bn * nvox << XSH);
      fg = (const unsigned char*)gp + ((long)bn * nvox << GSH);
    };
    u32x4 sx[NPIECE], sg[4];
    auto fetch_x = [&](int z, bool on) {
      const bool v = on && (unsigned)z < (unsigned)D;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)fx, 0, v ? (unsigned)(nvox << XSH) : 0u, 0x00020000);
      const int soff = v ? z * xplane_b : 0;
#pragma unroll
      for (int it = 0; it < NPIECE; ++it) {
        if constexpr (X16) { const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs, foff[it], soff, 0); sx[it] = (u32x4){h[0], h[1], 0u, 0u}; }
        else sx[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, foff[it], soff, 0);
      }
    };
    auto fetch_g = [&](int z0, bool on) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int z = z0 + (j >> 1);                               // pieces 0, 1: plane 0 of the tile; 2, 3: plane 1
        const bool v = on && z < D;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)fg, 0, v ? (unsigned)(nvox << GSH) : 0u, 0x00020000);
        const int soff = v ? z * gplane_b : 0;
        if constexpr (G16) { const u32x2 h = __builtin_amdgcn_raw_buffer_load_b64(rs, goff[j], soff, 0); sg[j] = (u32x4){h[0], h[1], 0u, 0u}; }
        else sg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, goff[j], soff, 0);
      }
    };
    auto to_bf16 = [&](const u32x4 v, bool is16) {
      if (is16) return __builtin_bit_cast(bf16x4t, (u32x2){v[0], v[1]});
      return __builtin_convertvector(__builtin_bit_cast(f32x4, v), bf16x4t);
    };
    auto commit_x = [&](int slot) {
      unsigned char* const dst = smem + slot * PLANE_B;
#pragma unroll
      for (int it = 0; it < 5; ++it) *(bf16x4t*)(dst + (frow0 + it) * (HXs * 32) + lane * 8) = to_bf16(sx[it], X16);
      if (e_ok) *(bf16x4t*)(dst + eldso) = to_bf16(sx[5], X16);
    };
    auto commit_g = [&](int gsel) {
      unsigned char* const dst = smem + TXP + gsel * TGB;
#pragma unroll
      for (int j = 0; j < 4; ++j) *(bf16x4t*)(dst + (ptid + 256 * j) * 8) = to_bf16(sg[j], G16);
    };
    // ---- operands.  lane (channel m, lane group k): record of voxel (row k >> 1, x = 4 (k & 1) + (m >> 2)) of a K group's two
    // rows, channel quad m & 3; the second transpose load of a K run sits 8 voxels (256 B) further along x
    const int a_lane = TXP + (((k >> 1) * WTX + 4 * (k & 1) + (m >> 2)) * 32) + (m & 3) * 8;      // + gsel*TGB + (z*8 + 2 rp) * 512
    const int b_lane = (((k >> 1) * HXs + 4 * (k & 1) + (m >> 2)) * 32) + (m & 3) * 8;            // + slot*PLANE_B + (2 rp + ky) * 576 + kx*32
    const int kz0 = (2 * wave) / 3, ky0 = (2 * wave) % 3, kz1 = (2 * wave + 1) / 3, ky1 = (2 * wave + 1) % 3;
    // Ten contraction steps per tile (K groups 0..7 for this wave's two stencil rows, then K groups 2w, 2w+1 for row 8);
    // the operands of step s + WG_TR_PF are read before the MFMAs of step s (pinned: the scheduler otherwise sinks a read to
    // its use and the wave waits out the LDS latency per operand)
    struct Step { bf16x8 a, b[6]; };
    auto contract = [&](int rot, int gsel) {
      const unsigned char* ab = smem + a_lane + gsel * TGB;
      const unsigned char* r0[2], *r1[2], *r8[2];
#pragma unroll
      for (int z = 0; z < 2; ++z) {
        r0[z] = smem + b_lane + bmod6(rot + z + kz0) * PLANE_B + ky0 * (HXs * 32);
        r1[z] = smem + b_lane + bmod6(rot + z + kz1) * PLANE_B + ky1 * (HXs * 32);
        r8[z] = smem + b_lane + bmod6(rot + z + 2) * PLANE_B + 2 * (HXs * 32);
      }
      Step st[WG_TR_PF + 1];
      auto load = [&](auto sc) {
        constexpr int sidx = decltype(sc)::v;
        Step& o = st[sidx % (WG_TR_PF + 1)];
        if constexpr (sidx < 8) {
          constexpr int z = sidx >> 2, rp = sidx & 3;
          const unsigned char* ap = ab + (z * WTY + 2 * rp) * (WTX * 32);
          o.a = tr_pair(ap, ap + 256);
          const unsigned char* p0 = r0[z] + 2 * rp * (HXs * 32);
          const unsigned char* p1 = r1[z] + 2 * rp * (HXs * 32);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            o.b[kx] = tr_pair(p0 + kx * 32, p0 + kx * 32 + 256);
            o.b[3 + kx] = tr_pair(p1 + kx * 32, p1 + kx * 32 + 256);
          }
        } else {
          const int kg = 2 * wave + (sidx - 8), z = kg >> 2, rp = kg & 3;      // (wave-uniform)
          const unsigned char* ap = ab + (z * WTY + 2 * rp) * (WTX * 32);
          o.a = tr_pair(ap, ap + 256);
          const unsigned char* p0 = (z ? r8[1] : r8[0]) + 2 * rp * (HXs * 32);
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) o.b[kx] = tr_pair(p0 + kx * 32, p0 + kx * 32 + 256);
        }
      };
      wstatic_for<0, WG_TR_PF>([&](auto sc) { load(sc); });
      wstatic_for<0, 10>([&](auto sc) {
        constexpr int sidx = decltype(sc)::v;
        if constexpr (sidx + WG_TR_PF < 10) load(WIC<sidx + WG_TR_PF>{});
        if constexpr (WG_TR_PF > 0) __builtin_amdgcn_sched_barrier(0);
        if constexpr (WG_TR_PF == 0) load(sc);
        const Step& o = st[sidx % (WG_TR_PF + 1)];
        if constexpr (sidx < 8) {
#pragma unroll
          for (int j = 0; j < 6; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.a, o.b[j], acc[j], 0, 0, 0);
        } else {
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[6 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o.a, o.b[j], acc[6 + j], 0, 0, 0);
        }
        if constexpr (WG_TR_PF > 0) __builtin_amdgcn_sched_barrier(0);
      });
    };

    int cx, cy, cz, cn;
    {
      int tt = t_begin;
      cz = tt % tiles_z; tt /= tiles_z;
      cx = tt % tiles_x; tt /= tiles_x;
      cy = tt % tiles_y; cn = tt / tiles_y;
    }
    auto step = [&](int& x_, int& y_, int& z_, int& n_) {         // the tile after (x_, y_, z_, n_): z fastest
      if (++z_ == tiles_z) { z_ = 0; if (++x_ == tiles_x) { x_ = 0; if (++y_ == tiles_y) { y_ = 0; ++n_; } } }
    };
    int rot = 0;
    if (producer) {
      // ---- waves 4..7: loads -> LDS, one tile ahead of the contraction.  Entering iteration t the incoming planes / gpre
      // block of tile t + 1 are in flight; they are committed at once (their slots / buffer were last read by tile t - 1),
      // then the loads of tile t + 2 are issued, then the workgroup barrier of tile t.
      auto prefetch = [&](int tx, int ty, int tz, int tn, bool on_, bool slide_) {
        if (on_ && !slide_) column(tx, ty, tn);
        if (!(WG_ABL & 4)) {
          fetch_x(tz * WTZ - 1 + (slide_ ? 2 : 0) + pzw, on_);
          fetch_g(tz * WTZ, on_);
        }
      };
      column(cx, cy, cn);
      fetch_x(cz * WTZ - 1 + pzw, true);
      fetch_g(cz * WTZ, true);
      commit_x(pzw);
      commit_g(0);
      fetch_x(cz * WTZ + 1 + pzw, true);
      commit_x(2 + pzw);
      lds_barrier_s();
      int ax = cx, ay = cy, az = cz, an = cn;                       // tile t + 1 (then t + 2)
      step(ax, ay, az, an);
      prefetch(ax, ay, az, an, t_begin + 1 < t_end, t_begin + 1 < t_end && az != 0);
      for (int t = t_begin; t < t_end; ++t) {
        const int gsel = (t - t_begin) & 1;
        const bool on = t + 1 < t_end;
        const int nz = az;                                          // (tile t + 1)
        const bool slide = on && nz != 0;
        if (on && !(WG_ABL & 2)) {
          commit_x(bmod6(rot + 4 + pzw));
          commit_g(gsel ^ 1);
        }
        step(ax, ay, az, an);                                       // -> tile t + 2
        const bool on2 = t + 2 < t_end, slide2 = on2 && az != 0;
        if (!on || slide) prefetch(ax, ay, az, an, on2, slide2);
        lds_barrier_s();
        if (on && !slide) {
          // tile t + 1 starts a column: what was committed are its planes 0, 1 (slots rot + 4, rot + 5); planes 2, 3 go to
          // the slots tile t has just released (exposed once per column), only then the loads of tile t + 2
          rot = bmod6(rot + 4);
          fetch_x(nz * WTZ + 1 + pzw, true);
          commit_x(bmod6(rot + 2 + pzw));
          lds_barrier_s();
          prefetch(ax, ay, az, an, on2, slide2);
        } else {
          rot = bmod6(rot + 2);
        }
      }
    } else {
      // ---- waves 0..3: the contraction of tile t while the other four waves stage tile t + 1
      lds_barrier_s();
      int az = cz;
      for (int t = t_begin; t < t_end; ++t) {
        const int gsel = (t - t_begin) & 1;
        const bool on = t + 1 < t_end;
        if (++az == tiles_z) az = 0;
        const bool slide = on && az != 0;
        if (!(WG_ABL & 1)) contract(rot, gsel);
        lds_barrier_s();
        if (on && !slide) {
          rot = bmod6(rot + 4);
          lds_barrier_s();
        } else {
          rot = bmod6(rot + 2);
        }
      }
    }
  }
  // row 8 (taps 24..26): waves 1..3 hand their shares to wave 0 through LDS, summed in wave order
  f32x4* red = (f32x4*)smem;
  __syncthreads();
  if (wave > 0 && wave < 4) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) red[((wave - 1) * 3 + kx) * 64 + lane] = acc[6 + kx];
  }
  __syncthreads();
  if (wave >= 4) return;                                          // (the staging waves hold no sums)
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; ++w)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) acc[6 + kx] += red[(w * 3 + kx) * 64 + lane];
  }
  float* out = partial + (long)blockIdx.x * 27 * 256;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r == 2 && wave != 0) break;
    const int zy = r == 2 ? 8 : 2 * wave + r;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < 4; ++i) out[(zy * 3 + kx) * 256 + (4 * k + i) * 16 + m] = acc[r * 3 + kx][i];
  }
}

// ---- bias gradient of a 16-channel layer: column sums of gpre [rows][16] ------------------------------
// A streaming reduction (the generic path runs it as a GEMM against a vector of ones with at most 512 blocks,
// 250 us for 134 MB; this takes ~40).  Fixed order: rows strided inside a block, blocks summed in fp64.
__global__ void __launch_bounds__(256) colsum16_partial_kernel(const f32x4* __restrict__ gp, float* __restrict__ partial,
                                                               long rows, int chunk) {
  const int t = threadIdx.x, q = t & 3, slot = t >> 2;
  const long r0 = (long)blockIdx.x * chunk, r1 = min(r0 + chunk, rows);
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long r = r0 + slot; r < r1; r += 64) acc += __builtin_nontemporal_load(gp + r * 4 + q);
  __shared__ f32x4 red[256];
  red[t] = acc;
  __syncthreads();
  if (t < 16) {
    float s = 0.f;
    for (int i = 0; i < 64; ++i) s += red[i * 4 + (t >> 2)][t & 3];
    partial[(long)blockIdx.x * 16 + t] = s;
  }
}

__global__ void __launch_bounds__(256) colsum16_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out,
                                                             float scale) {
  const int t = threadIdx.x, co = t & 15, grp = t >> 4;
  double s = 0.0;
  for (int b = grp; b < nblk; b += 16) s += (double)partial[(long)b * 16 + co];
  __shared__ double red[256];
  red[t] = s;
  __syncthreads();
  if (t < 16) {
    double tot = 0.0;
    for (int g = 0; g < 16; ++g) tot += red[g * 16 + t];
    out[t] = (float)(tot * (double)scale);
  }
}

struct WgradPlan { int taps, nct, ncit, chunk, nblk; };

bool wgrad_plan(int dims, long total, int Cin, int Cout, bool ones, WgradPlan& p) {
  if (dims != 0 && dims != 2 && dims != 3) return false;
  if (total <= 0 || total >= 0x7fffffffL || Cout <= 0 || (!ones && Cin <= 0)) return false;
  p.taps = dims == 3 ? 27 : (dims == 2 ? 9 : 1);
  p.nct = (Cout + 15) / 16;
  p.ncit = ones ? 1 : (Cin + 15) / 16;
  // enough blocks to fill the chip for small problems, at most ~512 partials per output for big ones
  long chunk = 1024;
  while ((total + chunk - 1) / chunk > 512) chunk *= 2;
  p.chunk = (int)chunk;
  p.nblk = (int)((total + chunk - 1) / chunk);
  return (long)p.nct * p.ncit <= 65535;
}

}  // namespace

static int wgrad_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cus;
}

static bool wgrad_fast3d(int dims, int N, int D, int H, int W, int Cin, int Cout) {
  return dims == 3 && Cin == 16 && Cout == 16 && (long)D * H * W * 64 < 0x7fffffffL && (long)N * D * H * W >= 8192;
}

extern "C" size_t lf_conv_bwd_weight_scratch_bytes(int dims, int N, int D, int H, int W, int Cin, int Cout) {
  if (wgrad_fast3d(dims, N, D, H, W, Cin, Cout)) return (size_t)wgrad_cus() * 8 * 27 * 256 * sizeof(float);
  WgradPlan p;
  if (!wgrad_plan(dims, (long)N * D * H * W, Cin > 0 ? Cin : 1, Cout, Cin <= 0, p)) return 0;
  return (size_t)p.nblk * p.taps * p.nct * p.ncit * 256 * sizeof(float);
}

extern "C" int lf_conv_bwd_weight(const float* x, const float* gpre, float* gw, void* scratch, size_t scratch_bytes,
                                  int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || gpre == nullptr || gw == nullptr) return LF_EINVAL;
  const bool ones = (x == nullptr);
  if (ones && Cout == 16 && lf_aligned16(gpre)) {
    const long rows = (long)N * D * H * W;
    long chunk = (rows + 1023) / 1024;                            // ~1024 blocks: four per CU keep enough loads in flight
    chunk = (chunk + 63) / 64 * 64;
    const int nblk = (int)((rows + chunk - 1) / chunk);
    if (scratch_bytes < (size_t)nblk * 16 * sizeof(float)) return LF_ENOSPC;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(colsum16_partial_kernel, dim3(nblk), dim3(256), 0, s, (const f32x4*)gpre, (float*)scratch, rows, (int)chunk);
    int st = lf_launch_status();
    if (st) return st;
    hipLaunchKernelGGL(colsum16_final_kernel, dim3(1), dim3(256), 0, s, (const float*)scratch, nblk, gw, scale);
    return lf_launch_status();
  }
  if (!ones && wgrad_fast3d(dims, N, D, H, W, Cin, Cout) && lf_aligned16(x) && lf_aligned16(gpre)) {
    const int cus = wgrad_cus();
    if (scratch_bytes < (size_t)cus * 8 * 27 * 256 * sizeof(float)) return LF_ENOSPC;
    const int ptx = (W + WTX - 1) / WTX, pty = (H + WTY - 1) / WTY, ptz = (D + WTZ - 1) / WTZ;
    const long pt = (long)ptx * pty * ptz * N;
    if (pt > 0x7fffffffL) return LF_EINVAL;
    const size_t shmem = (size_t)2 * WBUF;
    static lf_devmask_t attr_set;
    {
      hipError_t e = lf_ensure_dyn_lds(attr_set, (const void*)wgrad3d_c16_kernel, (int)shmem);
      if (e != hipSuccess) return (int)e;
    }
    hipStream_t s = (hipStream_t)stream;
    // every workgroup writes its 8 x 27 x 256 partials (zeros when it has no tiles), so the grid is always `cus`
    hipLaunchKernelGGL(wgrad3d_c16_kernel, dim3(cus), dim3(512), shmem, s, x, gpre, (float*)scratch, N, D, H, W, ptx, pty,
                       ptz, (int)pt);
    int st = lf_launch_status();
    if (st) return st;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(27, 1, 4), dim3(256), 0, s, (const float*)scratch, gw, cus, 27, 1, 1, 16, 16,
                       scale);
    return lf_launch_status();
  }
  if (ones) Cin = 1;
  WgradPlan p;
  if (!wgrad_plan(dims, (long)N * D * H * W, Cin, Cout, ones, p)) return LF_EINVAL;
  if (scratch_bytes < (size_t)p.nblk * p.taps * p.nct * p.ncit * 256 * sizeof(float)) return LF_ENOSPC;
  float* partial = (float*)scratch;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(p.nblk, p.taps, p.nct * p.ncit), block(256);
  if (dims == 3)
    hipLaunchKernelGGL((wgrad_partial_kernel<3>), grid, block, 0, s, x, gpre, partial, N, D, H, W, Cin, Cout, p.chunk, p.ncit);
  else if (dims == 2)
    hipLaunchKernelGGL((wgrad_partial_kernel<2>), grid, block, 0, s, x, gpre, partial, N, D, H, W, Cin, Cout, p.chunk, p.ncit);
  else
    hipLaunchKernelGGL((wgrad_partial_kernel<0>), grid, block, 0, s, x, gpre, partial, N, D, H, W, Cin, Cout, p.chunk, p.ncit);
  int st = lf_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(p.taps, p.nct * p.ncit, 4), block, 0, s, partial, gw, p.nblk, p.taps,
                     p.nct * p.ncit, p.ncit, Cin, Cout, scale);
  return lf_launch_status();
}

static int wgrad_bf16_launch(const void* x, const void* gpre, float* gw, void* scratch, size_t scratch_bytes,
                             int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, int io, void* stream);

extern "C" int lf_conv_bwd_weight_bf16(const float* x, const float* gpre, float* gw, void* scratch, size_t scratch_bytes,
                                       int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, void* stream) {
  return wgrad_bf16_launch(x, gpre, gw, scratch, scratch_bytes, dims, N, D, H, W, Cin, Cout, scale, 0, stream);
}

extern "C" int lf_conv_bwd_weight_bf16_io(const void* x, const void* gpre, float* gw, void* scratch, size_t scratch_bytes,
                                          int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, int io, void* stream) {
  if (io < 0 || io > 3) return LF_EINVAL;
  return wgrad_bf16_launch(x, gpre, gw, scratch, scratch_bytes, dims, N, D, H, W, Cin, Cout, scale, io, stream);
}

static int wgrad_bf16_launch(const void* x, const void* gpre, float* gw, void* scratch, size_t scratch_bytes,
                             int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, int io, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || x == nullptr || gpre == nullptr || gw == nullptr) return LF_EINVAL;
  if (!wgrad_fast3d(dims, N, D, H, W, Cin, Cout) || !lf_aligned16(x) || !lf_aligned16(gpre)) return LF_EINVAL;
  // the kernel's 32-bit offsets reach three planes past the sample (halo planes of the last tile + the pair's second plane)
  if ((long)(D + 3) * H * W * 64 > 0xffffffffL) return LF_EINVAL;
#ifdef WG_OLD
  const int nb = 2 * wgrad_cus();
#else
  const int nb = WG_TR_WGS * wgrad_cus();
#endif
  if (scratch_bytes < (size_t)nb * 27 * 256 * sizeof(float)) return LF_ENOSPC;
  const int ptx = (W + WTX - 1) / WTX, pty = (H + WTY - 1) / WTY, ptz = (D + WTZ - 1) / WTZ;
  const long pt = (long)ptx * pty * ptz * N;
  if (pt > 0x7fffffffL) return LF_EINVAL;
  typedef void (*kern_t)(const float*, const float*, float*, int, int, int, int, int, int, int, int);
#ifdef WG_OLD                                                      // (A/B: the channel-major kernel of rounds 2-5, tools/wgrad_ab.py)
  const size_t shmem = (size_t)BLDS;
  static const kern_t kerns[4] = {wgrad3d_c16_bf16_kernel<0>, wgrad3d_c16_bf16_kernel<1>, wgrad3d_c16_bf16_kernel<2>, wgrad3d_c16_bf16_kernel<3>};
#else
  const size_t shmem = (size_t)TLDS;
  static const kern_t kerns[4] = {wgrad3d_c16_tr_kernel<0>, wgrad3d_c16_tr_kernel<1>, wgrad3d_c16_tr_kernel<2>, wgrad3d_c16_tr_kernel<3>};
#endif
  static lf_devmask_t attr_set[4];
  {
    hipError_t e = lf_ensure_dyn_lds(attr_set[io], (const void*)kerns[io], (int)shmem);
    if (e != hipSuccess) return (int)e;
  }
  hipStream_t s = (hipStream_t)stream;
#ifdef WG_OLD
  const unsigned threads = 256;
#else
  const unsigned threads = 512;                                   // four contraction waves + four staging waves
#endif
  hipLaunchKernelGGL(kerns[io], dim3(nb), dim3(threads), shmem, s, (const float*)x, (const float*)gpre, (float*)scratch, N, D, H, W, ptx, pty, ptz, (int)pt);
  int st = lf_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(27, 1, 4), dim3(256), 0, s, (const float*)scratch, gw, nb, 27, 1, 1, 16, 16, scale);
  return lf_launch_status();
}
