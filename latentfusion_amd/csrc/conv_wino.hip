// Winograd F(2x2x2, 3x3x3) form of the fused 16 -> 16 channel conv3d block step (gfx950).
//
//   y = PixelNorm(LeakyReLU(conv3d(x, W) * he + b))            latentfusion/modules/blocks.py:152-158
//                                                              latentfusion/modules/equalized.py:57-64
// and, with transposed/flipped weights and `prev_*` set, the data gradient fused with the previous
// layer's LeakyReLU'/PixelNorm' (autograd of the same lines).
//
// All arithmetic is fp32 (v_mfma_f32_16x16x4_f32 + fp32 VALU transforms).  The minimal-filtering
// identity  Y = A^T[(G w G^T) . (B^T d B)]A  applied along z, y and x needs 64 multiplies per 2x2x2
// outputs instead of 216, so the MFMA work per voxel drops 3.375x; measured against fp64 the result is
// as close as the direct fp32 kernel's (tests/test_ops_gpu.py::test_winograd_conv3d_*).
//
// Work split: 256-thread workgroups, TWO per CU (2 x 79,872 B of LDS), each persistent over its own range
// of 2 x 8 x 16-voxel tiles.  The two workgroups of a CU are not synchronised with each other, so one's
// exchange / epilogue / DMA-wait phases run under the other's MFMA phase (on gfx950 the fp32 MFMA and the
// fp32 VALU share the SIMD's issue pipe -- measured: they do not overlap -- so everything that is not an
// MFMA has to be either few instructions or hidden this way).
//   * wave w owns z-frequency a = w of the tile; the 16 (y,x)-frequency matrices U[a][b][c]
//     (16 cout x 16 cin each) live in 64 VGPRs for the whole launch;
//   * MFMA columns = 16 Winograd tiles (2 in y x 8 in x), lane group kg = lane >> 4 carries input
//     channels 4kg..4kg+3 (B operand) and receives output channels 4kg..4kg+3 (D operand);
//   * per 16-tile group: 32 ds_read_b128 of the fp32 halo (two z planes combined on the fly),
//     x/y input transforms in registers, 64 MFMAs, y/x output transform in registers;
//   * the four z-frequency partials of a tile meet through LDS, then every wave finishes a quarter of the
//     outputs: z output transform, He scale, bias, LeakyReLU, PixelNorm (DPP quad reduction), and one
//     x-contiguous 1 KiB row per store instruction.
// The halo (4 x 10 x 18 voxels, fp32) slides along z: a workgroup walks up a column of tiles, moves the upper two
// planes of the halo LDS -> LDS and fetches only the two new planes (22.5 KiB) per tile, staged through registers
// under the exchange / epilogue phase.  LDS layout: voxel slot = (z*10 + y)*18 + (x&1)*9 + (x>>1), 16-byte quarter
// q stored at q ^ s, s = 2*((slot>>2)&1) + ((y>>1)&1)  ->  every ds_read_b128 lane group hits 16 distinct slots
// (SQ_LDS_BANK_CONFLICT = 0).
#include "lf_common.h"
#ifndef WINO_ABL
#define WINO_ABL 0      // ablations for tools/wino_ab.py (results are WRONG, timings only): 1 = no epilogue arithmetic (upper bound on
#endif                  // what moving the epilogue to other waves could give: they would still issue it on the shared pipe);
                        // 2 = no z-frequency exchange / z output transform (each wave stores from its own partials); 4 = every MFMA
                        // chain 1.5x as long (the MFMA count of F(2,3) in y,x with direct z taps); 16 / 32 = cycle stamps; 64 = one LDS read per
                        // value and no z combination (upper bound on staging z-combined planes)
// wave priorities (s_setprio) of the three phases of a tile: halo reads + input transforms (latency-bound: LDS),
// the MFMA chains (throughput-bound) and the exchange / epilogue / halo fetch (latency-bound: LDS, HBM)
#ifndef WINO_PJ_ABL
#define WINO_PJ_ABL 0   // ablations of the fused projection backward (tools/proj_fuse_ab.py --variants; results WRONG, timings only):
#endif                  // 1 = no activation / norm loads (the HBM streams of the halo phase), 2 = no MFMAs / epilogue arithmetic,
                        // 4 = no slide copy
#ifndef WINO_PL
#define WINO_PL 2
#endif
#ifndef WINO_PM
#define WINO_PM 1
#endif
#ifndef WINO_PN
#define WINO_PN 3
#endif

namespace {

constexpr int TZw = 2, TYw = 8, TXw = 16;
constexpr int HZw = TZw + 2, HYw = TYw + 2, HXw = TXw + 2;     // 4 x 10 x 18 halo
constexpr int HALOw = HZw * HYw * HXw;                          // 720 voxels
constexpr int HALFw = 2 * HYw * HXw;                            // 360 voxels: two z planes of the halo
constexpr int HALFBw = HALFw * 64;                              // 23,040 B
constexpr int NSLOTw = (HALFw * 4 + 63) / 64;                   // 23 pieces of 1 KiB per half (the last one half full)
constexpr int NITw = (NSLOTw + 3) / 4;                          // 6 pieces per wave and half
constexpr int BUFw = 2 * HALFBw;                                // 46,080 B halo buffer
constexpr int PXw = 4 * 8192;                                   // 32,768 B partial-exchange region
constexpr int LDSw = BUFw + PXw + 1024;                         // + 1 KiB DMA scratch = 79,872 B (two workgroups per CU)

__device__ __forceinline__ void lds_barrier() {
  // all LDS traffic of this wave retired, then workgroup barrier; global loads/stores and the LDS-DMA of
  // the next tile stay in flight (a __syncthreads() would drain vmcnt as well)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// sum over the four lanes of a quad (4q .. 4q+3) with DPP quad_perm, same value in all four
__device__ __forceinline__ float quad_sum(float v) {
  const float a = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // [1,0,3,2]
  return a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xF, 0xF, true));            // [2,3,0,1]
}

// v_rsq_f32 / v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float fast_rsqrt(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  return r * (1.5f - 0.5f * x * r * r);
}
__device__ __forceinline__ float fast_rcp(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return r * (2.f - x * r);
}

// a - b as two v_pk_add_f32 with negated second source.  (LLVM packs <4 x float> fadd into v_pk_add_f32 but
// scalarises fsub; on this kernel every VALU instruction competes with the fp32 MFMAs for the issue pipe.)
// Only used where neither operand is an MFMA result and the result does not feed an MFMA directly: the
// MFMA <-> VALU wait states are inserted by the compiler, which does not look inside asm (feeding the
// B operand of an MFMA straight from this asm gave wrong results on gfx950).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 pk_sub(f32x4 a, f32x4 b) {
  f32x2 lo, hi;
  const f32x2 alo = {a[0], a[1]}, ahi = {a[2], a[3]}, blo = {b[0], b[1]}, bhi = {b[2], b[3]};
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(alo), "v"(blo));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(ahi), "v"(bhi));
  return (f32x4){lo[0], lo[1], hi[0], hi[1]};
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// SPLIT = false: all-fp32 products, wt = U[a][(b,c)][k-chunk] for v_mfma_f32_16x16x4_f32.
// SPLIT = true : every Winograd-domain product U.V from three v_mfma_f32_16x16x16_f16 accumulating in fp32,
//                U_hi.V_hi + U_hi.V_lo + U_lo.V_hi  (x = x_hi + x_lo, both f16; dropped term <= 2^-22 |U||V|);
//                whi / wlo = the two halves of U, V is split on the fly after scaling by the power of two in_scale.
// OFFX (the fused forms, which have no registers to spare): only off[r][0] is kept; the rows dy = 2, 3, whose quarter
// swizzle differs in bit 0, derive their offset with one v_xor per read instead of a second register per residue.
template <int A, bool SPLIT, bool OFFX = false>
__device__ __forceinline__ void wino_compute(const unsigned char* __restrict__ buf, const int (&off)[8][2],
                                              const float (&wt)[64], const f16x4 (&whi)[16], const f16x4 (&wlo)[16],
                                              float in_scale, unsigned char* __restrict__ pdst,
                                              const int (&pw)[2], unsigned* tsp = nullptr) {
#define CTS(k) do { if ((WINO_ABL & 32) && tsp != nullptr) tsp[k] = (unsigned)__builtin_readcyclecounter(); } while (0)
  // z input transform of frequency A: d0-d2, d1+d2, d2-d1, d1-d3
  constexpr int DZ0 = (A == 0) ? 0 : (A == 2 ? 2 : 1);
  constexpr int DZ1 = (A == 0) ? 2 : (A == 1 ? 2 : (A == 2 ? 1 : 3));
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    f32x4 Yg[4];                                              // (j, i) outputs of this group, z-frequency A
    f32x4 vx[4][4];                                           // [dy][x-frequency c]
    // the halo is read half a row (2 z planes x 2 x positions = 4 ds_read_b128) ahead of the half row being
    // transformed; the fences stop the scheduler from folding this back into load-wait-use pairs, which
    // exposes the full LDS latency sixteen times per group
    f32x4 raw[2][4];
    auto load_half = [&](int hh, f32x4 (&r)[4]) {
      const int dy = hh >> 1;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int dx = (hh & 1) * 2 + e;
        const int ca = ((DZ0 * 10 + 4 * g + dy) * 18 + (dx & 1) * 9 + (dx >> 1));
        const int cb = ((DZ1 * 10 + 4 * g + dy) * 18 + (dx & 1) * 9 + (dx >> 1));
        if constexpr ((WINO_ABL & 64) != 0) {                  // ablation: ONE plane read per value, no z combination (an upper
          r[2 * e] = *(const f32x4*)(buf + off[ca & 7][dy >> 1] + ca * 64);   // bound on staging z-COMBINED planes in LDS)
          r[2 * e + 1] = r[2 * e];
        } else if constexpr (OFFX) {
          int oa = off[ca & 7][0], ob = off[cb & 7][0];
          if (dy >> 1) {
            asm volatile("" : "+v"(oa), "+v"(ob));               // (opaque: or the xor-ed copies are hoisted back into registers)
            oa ^= 16;
            ob ^= 16;
          }
          r[2 * e] = *(const f32x4*)(buf + oa + ca * 64);
          r[2 * e + 1] = *(const f32x4*)(buf + ob + cb * 64);
        } else {
          r[2 * e] = *(const f32x4*)(buf + off[ca & 7][dy >> 1] + ca * 64);
          r[2 * e + 1] = *(const f32x4*)(buf + off[cb & 7][dy >> 1] + cb * 64);
        }
      }
    };
    __builtin_amdgcn_s_setprio(WINO_PL);
    load_half(0, raw[0]);
    f32x4 d[4];
#pragma unroll
    for (int hh = 0; hh < 8; ++hh) {
      if (hh < 7) load_half(hh + 1, raw[(hh + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
      const int dy = hh >> 1;
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const f32x4 va = raw[hh & 1][2 * e], vb = raw[hh & 1][2 * e + 1];
        if constexpr ((WINO_ABL & 64) != 0) d[(hh & 1) * 2 + e] = va;
        else d[(hh & 1) * 2 + e] = (A == 1) ? (va + vb) : pk_sub(va, vb);
      }
      if (hh & 1) {
        vx[dy][0] = pk_sub(d[0], d[2]);
        vx[dy][1] = d[1] + d[2];
        vx[dy][2] = pk_sub(d[2], d[1]);
        vx[dy][3] = pk_sub(d[1], d[3]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    CTS(9 + 2 * g);
    __builtin_amdgcn_s_setprio(WINO_PM);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x4 m[4];
      // two frequencies of the row advance together, one k-chunk at a time: a 16x16x4 f32 MFMA can issue every
      // 32 cycles but its result feeds a dependent one only after 40, so a single back-to-back chain would bubble
#pragma unroll
      for (int cp = 0; cp < 4; cp += 2) {
        f32x4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int c = cp + h;
          v[h] = (b == 0) ? (vx[0][c] - vx[2][c])
               : (b == 1) ? (vx[1][c] + vx[2][c])
               : (b == 2) ? (vx[2][c] - vx[1][c])
                          : (vx[1][c] - vx[3][c]);
          m[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (!SPLIT) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
              m[cp + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[(b * 4 + cp + h) * 4 + i], v[h][i], m[cp + h], 0, 0, 0);
          if constexpr ((WINO_ABL & 4) != 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int h = 0; h < 2; ++h)
                m[cp + h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wt[(b * 4 + cp + h) * 4 + i], v[h][i], m[cp + h], 0, 0, 0);
          }
        } else {
          // fp32 Winograd-domain value -> f16 hi + lo (exact to 22 bits), three product terms
          f16x4 vhi[2], vlo[2], vl2[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f32x4 vs = v[h] * in_scale;
            vhi[h] = __builtin_convertvector(vs, f16x4);
            const f32x4 r1 = vs - __builtin_convertvector(vhi[h], f32x4);
            vlo[h] = __builtin_convertvector(r1, f16x4);
            if constexpr ((WINO_ABL & 8) != 0)                  // timing proxy of a THREE-piece split: third piece + 6 products
              vl2[h] = __builtin_convertvector(r1 - __builtin_convertvector(vlo[h], f32x4), f16x4);
          }
#pragma unroll
          for (int h = 0; h < 2; ++h) m[cp + h] = __builtin_amdgcn_mfma_f32_16x16x16f16(whi[b * 4 + cp + h], vlo[h], m[cp + h], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < 2; ++h) m[cp + h] = __builtin_amdgcn_mfma_f32_16x16x16f16(wlo[b * 4 + cp + h], vhi[h], m[cp + h], 0, 0, 0);
#pragma unroll
          for (int h = 0; h < 2; ++h) m[cp + h] = __builtin_amdgcn_mfma_f32_16x16x16f16(whi[b * 4 + cp + h], vhi[h], m[cp + h], 0, 0, 0);
          if constexpr ((WINO_ABL & 8) != 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h) m[cp + h] = __builtin_amdgcn_mfma_f32_16x16x16f16(wlo[b * 4 + cp + h], vlo[h], m[cp + h], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) m[cp + h] = __builtin_amdgcn_mfma_f32_16x16x16f16(whi[b * 4 + cp + h], vl2[h], m[cp + h], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < 2; ++h) m[cp + h] = __builtin_amdgcn_mfma_f32_16x16x16f16(wlo[b * 4 + cp + h], vl2[h], m[cp + h], 0, 0, 0);
          }
        }
      }
      // x output transform, then accumulate the y output transform
      const f32x4 t0 = m[0] + m[1] + m[2];
      const f32x4 t1 = m[1] - m[2] - m[3];
      if (b == 0) { Yg[0] = t0; Yg[1] = t1; }
      if (b == 1) { Yg[0] += t0; Yg[1] += t1; Yg[2] = t0; Yg[3] = t1; }
      if (b == 2) { Yg[0] += t0; Yg[1] += t1; Yg[2] = pk_sub(Yg[2], t0); Yg[3] = pk_sub(Yg[3], t1); }
      if (b == 3) { Yg[2] = pk_sub(Yg[2], t0); Yg[3] = pk_sub(Yg[3], t1); }
    }
    // partial outputs of z-frequency A -> the exchange region (separate from the halo, free since the last
    // barrier of the previous tile): 1 KiB blocks [wave][g][tyb][j], 64 float4 slots each at (L ^ ((L >> 3) & 7)),
    // L = x*4 + channel quarter -- the writers' 8-lane groups and the readers' 16-lane groups both hit distinct
    // bank slots, and a reader wave gets one x-contiguous 1 KiB output row per instruction
#pragma unroll
    for (int ji = 0; ji < 4; ++ji) *(f32x4*)(pdst + g * 4096 + (ji >> 1) * 1024 + pw[ji & 1]) = Yg[ji];
    CTS(10 + 2 * g);
  }
#undef CTS
}

// The renderer's factor projection (reference modules/geometry.py:731-749: the depth axis folded into the channels of a
// 1x1 convolution, C*D -> 16) sits directly behind the last camera block and contracts exactly the axis a workgroup of
// this kernel walks (a column of tiles along z).  Two fused forms remove its HBM round trips (round 4):
//   FUSE = 1 (forward): the finished output records of a tile (after LeakyReLU / PixelNorm) are also contracted with the
//     projection weights of their two depth planes into a per-wave 2 rows x 16 x x 16 cout accumulator (16 extra MFMAs per
//     wave and tile; the records cross from the epilogue's lane = (x, quarter) order to the MFMA's B-operand order
//     through the wave's own, already consumed slices of the exchange region); at the top of the column the projection's
//     own epilogue (He, bias, LeakyReLU, PixelNorm) runs and the (H, W, 16) latent image rows are stored.  Same
//     operands in the same accumulation order as lf_conv1x1_fwd on the stored volume: bit-identical.
//   FUSE = 2 (backward): the input gradient volume of the first data-gradient convolution never exists in HBM.  It is
//     g(z, y, x, :) = LReLU'/PixelNorm'[ he_p * Wt_p(z) . gp(y, x, :) ] with the saved activation / norm of the last
//     camera block; a workgroup forms the two new z planes of its halo per tile from the block's activation record
//     (same bytes the gradient volume would have cost), the 16-float projection gradient of the pixel (resident in
//     registers for the whole column) and the plane's 16 x 16 weight slice: 24 extra MFMAs per wave and tile instead of a
//     2.1 GB kernel.  Same arithmetic as lf_conv1x1_bwd_data's epilogue.
struct WinoProj {
  const float* wA;        // FUSE 1: [D][64 lanes][4] = Wp[cout = lane & 15][d][c = 4 (lane >> 4) + i]; FUSE 2: the transposed slices
  const float* bias;      // FUSE 1: projection bias (16) or null
  float* zp;              // FUSE 1: out, (N, H, W, 16) projected latent image
  float* pnorm;           // FUSE 1: out, (N * H * W) PixelNorm denominators of zp (or null)
  const float* gp;        // FUSE 2: in, (N, H, W, 16) gradient w.r.t. the projection's pre-activation
  const float* xnorm;     // FUSE 2: in, (N * D * H * W) PixelNorm denominators saved with the activation passed as `x`
  float he;               // the projection's He constant
  unsigned flags;         // FUSE 1: the projection's epilogue; FUSE 2: the epilogue of the layer that produced `x`
};

// 2-bit table p = (0, 2, 3, 1): a 16-byte quarter q of record v (16 records of 64 B) stored at position q ^ p(v >> 2) is read
// conflict-free by ds_read_b128 with lane = q * 16 + v (the hardware's lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...)
__device__ __forceinline__ int tr_swz(int v) { return (0x78 >> (2 * ((v >> 2) & 3))) & 3; }

template <bool SPLIT, int FUSE = 0>
__device__ __forceinline__ void conv3d_c16_wino_body(
    const float* __restrict__ x, const float* __restrict__ upack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z, int ntiles,
    float he, unsigned flags, float slope, float eps,
    const float* __restrict__ prev_y, const float* __restrict__ prev_norm, unsigned prev_flags,
    const float* __restrict__ amax_in, float* __restrict__ amax_out, const WinoProj pj = WinoProj()) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const buf = smem;                              // halo
  unsigned char* const px = smem + BUFw;                        // partial exchange

  const int tid = threadIdx.x, lane = tid & 63;
  const int fa = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave = z-frequency
  const int n = lane & 15, kg = lane >> 4;
  const int tx = n & 7, tyb = n >> 3;

  // workgroups b, b+8, b+16, ... run on the same XCD (one L2 each): give them consecutive tile ranges so
  // that the z-halo planes shared by neighbouring ranges are fetched from HBM once
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
  // (the fused forms keep per-column state: their ranges are whole columns of tiles)
  const int per = (FUSE != 0) ? ((ntiles / tiles_z + nb - 1) / nb) * tiles_z : (ntiles + nb - 1) / nb;
  const int t_begin = lb * per;
  const int t_end = min(t_begin + per, ntiles);
  if (t_begin >= t_end) return;

  const long nvox = (long)D * H * W;
  const unsigned sample_bytes = (unsigned)(nvox * 64);

  // ---- LDS-DMA piece constants: LDS slot p = piece*64 + lane holds quarter (p&3)^s of voxel slot p>>2 ----
  // ---- halo piece constants: within a half (two z planes), LDS slot p = piece*64 + lane holds quarter (p&3)^s of
  // voxel slot p>>2.  (360 slots per half is a multiple of 8, so the swizzle is the same in both halves.)
  // Per piece a lane keeps only its byte offset relative to the half's first voxel; x / y range violations (possible
  // only in the first / last tile of a row or column of tiles) are five flag bits per piece in one register, z range
  // violations fall outside the buffer descriptor's range on their own and read zeros. ----
  enum { F_XLO = 1, F_XHI = 2, F_YLO = 4, F_YHI = 8, F_PAD = 16, F_ALL6 = 0x2108421 };
  const int xlim = W - (tiles_x - 1) * TXw + 1, ylim = H - (tiles_y - 1) * TYw + 1;   // first invalid lx / ly in the last tile
  int hrel[NITw];
  unsigned hflags = 0;
#pragma unroll
  for (int it = 0; it < NITw; ++it) {
    const int p = (fa + 4 * it) * 64 + lane;
    const int vs = p >> 2;
    const int row = vs / HXw, rem = vs - row * HXw;
    const int xl = rem / 9, xa = rem - xl * 9;
    const int lx = 2 * xa + xl, ly = row % HYw, lz = row / HYw;
    const int sw = (((vs >> 2) & 1) << 1) | ((ly >> 1) & 1);
    const int q = (p & 3) ^ sw;
    const bool pad = vs >= HALFw;
    hrel[it] = pad ? 0 : ((lz * H + ly) * W + lx) * 64 + q * 16;
    const unsigned f = pad ? F_PAD : ((lx == 0 ? F_XLO : 0) | (lx >= xlim ? F_XHI : 0) | (ly == 0 ? F_YLO : 0) | (ly >= ylim ? F_YHI : 0));
    hflags |= f << (5 * it);
  }
  const bool last_piece_ok = ((fa + 4 * (NITw - 1)) * 64 + lane) < HALFw * 4;      // piece 22: lanes 0-31; 23: none

  // ---- B-operand addressing: byte offset of (lane's tile, channel quarter) for the 8 slot residues ----
  // (rows dy = 2,3 sit one (y>>1) step further: their quarter swizzle differs in bit 0)
  int off[8][2];
#pragma unroll
  for (int r = 0; r < 8; ++r)
#pragma unroll
    for (int dyb = 0; dyb < 2; ++dyb) {
      const int bit = ((36 * tyb + tx + r) >> 2) & 1;
      const int quarter = (kg ^ tyb ^ dyb) ^ (bit << 1);
      off[r][dyb] = (36 * tyb + tx) * 64 + quarter * 16;
    }

  // ---- partial-exchange addressing ----
  int pw[2];                                                   // writer: lane part of the slot, per x parity i
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int L = (2 * tx + i) * 4 + kg;
    pw[i] = tyb * 2048 + (L ^ ((L >> 3) & 7)) * 16;
  }
  const int pr = (lane ^ ((lane >> 3) & 7)) * 16;              // reader: lane = x*4 + quarter
  const int ex = lane >> 2, eq = lane & 3;

  // ---- transformed weights of this wave's z-frequency, lane-major in memory: fp32 [16 (b,c)][4 k-chunks], or
  // f16 [16 (b,c)][hi, lo] x 4 cin per lane (A operands, K = 4 cin per lane group) ----
  float wt[64];
  f16x4 whi[16], wlo[16];
  float in_scale = 1.f;
  if constexpr (!SPLIT) {
    const float* up = upack + (long)fa * 64 * 64 + lane;
#pragma unroll
    for (int k = 0; k < 64; ++k) wt[k] = up[k * 64];
  } else {
    const f16x4* up = (const f16x4*)upack + (long)fa * 16 * 2 * 64 + lane;
#pragma unroll
    for (int k = 0; k < 16; ++k) { whi[k] = up[(k * 2 + 0) * 64]; wlo[k] = up[(k * 2 + 1) * 64]; }
    // power-of-two input scale from the tensor's max-abs (gradient launches); 1 otherwise
    if (amax_in != nullptr) {
      const float am = lf_amax_read(amax_in, lane);
      if (am > 0.f && am < 3.0e38f) {
        int ex;
        frexpf(am, &ex);                                          // am = m * 2^ex, m in [0.5, 1)
        in_scale = ldexpf(1.f, 9 - ex);                           // max maps into [2^8, 2^9): x8 transform growth stays in f16 range
      }
    }
  }

  // tile coordinates are stepped, not divided: (cx, cy, cz, cn) = tile t, (nx, ny, nz, nn) = tile t + 1
  int cx, cy, cz, cn;
  {
    int tt = t_begin;                                            // z fastest: a workgroup walks up columns of tiles
    cz = tt % tiles_z; tt /= tiles_z;
    cx = tt % tiles_x; tt /= tiles_x;
    cy = tt % tiles_y; cn = tt / tiles_y;
  }
  // The halo slides along z: the upper two planes of a tile's halo are the lower two of the next tile up the
  // column, so they are moved LDS -> LDS and only two new planes (22.5 KiB) are fetched per tile; at the bottom of a
  // column all four planes are fetched.  Everything is staged through registers: requested right after the
  // barrier that frees the halo buffer, in flight under the exchange / epilogue arithmetic, written to LDS
  // (linear: piece * 1 KiB + lane * 16 within a half) just before the tile's last barrier.  (The vector-memory
  // path of a CU sustains only ~16 B/clk on this access pattern and blocks the issuing wave while its queue is
  // full, so halving the bytes shortens the non-MFMA phase of every tile.)
  u32x4 hlo[NITw], hhi[NITw];
  auto fetch_half = [&](u32x4 (&dst)[NITw], __amdgpu_buffer_rsrc_t rs, int base, unsigned sel) {
    // (opaque to the optimiser, or base + hrel[] of every branch is hoisted out of the tile loop into registers)
#pragma unroll
    for (int i = 0; i < NITw; ++i) asm volatile("" : "+v"(hrel[i]));
    if (sel == 0) {                                             // wave-uniform: interior tile, one VALU add per piece
#pragma unroll
      for (int it = 0; it < NITw; ++it) dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + hrel[it], 0, 0);
    } else {
      const unsigned bad = hflags & sel;
#pragma unroll
      for (int it = 0; it < NITw; ++it)
        dst[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, (bad & (31u << (5 * it))) ? 0x7fffffff : base + hrel[it], 0, 0);
    }
  };
  auto halo_fetch = [&](int bx, int by, int bz, int bn, bool on, bool slide) {
    // everything wave-uniform here runs on the scalar unit
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)(on ? bn : 0) * nvox * 16), 0,
                                                                        on ? sample_bytes : 0u, 0x00020000);
    const int base = (((bz * TZw - 1) * H + (by * TYw - 1)) * W + (bx * TXw - 1)) * 64;
    unsigned sel = 0;
    if (bx == 0) sel |= F_XLO * F_ALL6;
    if (bx == tiles_x - 1) sel |= F_XHI * F_ALL6;
    if (by == 0) sel |= F_YLO * F_ALL6;
    if (by == tiles_y - 1) sel |= F_YHI * F_ALL6;
    if (slide) {
#pragma unroll
      for (int it = 0; it < NITw; ++it) hlo[it] = *(const u32x4*)(buf + HALFBw + (fa + 4 * it) * 1024 + lane * 16);
    } else {
      fetch_half(hlo, rs, base, sel);
    }
    fetch_half(hhi, rs, base + 2 * H * W * 64, sel);
  };
  auto halo_commit = [&]() {
#pragma unroll
    for (int it = 0; it < NITw; ++it) {
      unsigned char* dst = buf + (fa + 4 * it) * 1024 + lane * 16;
      if (it < NITw - 1 || last_piece_ok) {
        *(u32x4*)dst = hlo[it];
        *(u32x4*)(dst + HALFBw) = hhi[it];
      }
    }
  };
  const float out_scale = he / in_scale;
  const bool addmode = prev_y != nullptr && (prev_flags & LF_EPI_ADD);      // prev_y is an addend, not a saved activation
  float wave_amax = 0.f;

  // ---- FUSE = 2: the halo is COMPUTED, not fetched.  A z plane of the halo (180 voxel slots) is cut into 12 groups of 16
  // consecutive slots (the last one holds 4); wave fa owns groups fa, fa + 4, fa + 8 of both planes of a half: lane =
  // kg * 16 + n is (channel quarter kg, slot 16 * group + n) -- the D-operand order of the MFMA that forms the record and
  // the order of lf_conv1x1_bwd_data's epilogue.  Every LDS byte of a group region (1 KiB per plane) is read (slide) and
  // written by its owner wave only, so the slide needs no barrier of its own. ----
  int pj_lo[3];                                                // LDS byte offset of (slot, quarter) in plane 0 of a half
  int pj_pix[3];                                               // per column: pixel index gy * W + gx of the slot, -1 = outside
  f32x4 pj_b[3];                                               // per tile: B operands = gp(pixel, 4 kg .. 4 kg + 3)
  f32x4 pj_aw[2], pj_yp[6];
  float pj_nr[6];
  const bool pj_last_ok = n < 4;                               // group 11 = slots 176 .. 179
  auto pj_column = [&](int bx, int by, int bn) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int sl = (fa + 4 * it) * 16 + n;
      const int ly = sl / HXw, rem = sl - ly * HXw;
      const int xl = rem / 9, xa = rem - xl * 9;
      const int gy = by * TYw - 1 + ly, gx = bx * TXw - 1 + 2 * xa + xl;
      const bool ok = sl < HYw * HXw && gy >= 0 && gy < H && gx >= 0 && gx < W;
      pj_pix[it] = ok ? gy * W + gx : -1;
    }
  };
  // the loads of a half in two parts: the HBM streams (activation records, norms) are requested early, under the
  // exchange / epilogue arithmetic; the L2-resident operands (the pixels' gradient records, the planes' weight slices)
  // late, behind that arithmetic, when its registers are free
  auto pj_issue_hot = [&](int z0, int bn) {
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(pj_pix[i]));
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(pj.gp + (long)bn * H * W * 16), 0,
                                                                        (unsigned)(H * W * 64), 0x00020000);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rg, pj_pix[it] >= 0 ? pj_pix[it] * 64 + kg * 16 : 0x7fffffff, 0, 0);
      pj_b[it] = (f32x4){__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3])};
    }
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int gz = z0 + pl;                                    // wave-uniform
      const bool pok = gz >= 0 && gz < D;
      pj_aw[pl] = *(const f32x4*)(pj.wA + ((long)(pok ? gz : 0) * 64 + lane) * 4);
      if (!pok) pj_aw[pl] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto pj_issue = [&](int z0, int bn) {
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)bn * nvox * 16), 0, sample_bytes, 0x00020000);
    const bool has_norm = pj.xnorm != nullptr;                   // (wave-uniform)
    const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)(has_norm ? pj.xnorm + (long)bn * nvox : x), 0,
                                                                        has_norm ? (unsigned)(nvox * 4) : 0u, 0x00020000);
    // (the pixels' gradient records are re-read per tile -- 11.5 KB per workgroup, L1 / L2 hits -- rather than held in 12
    // registers across the MFMA phase, where the kernel has none to spare)
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(pj_pix[i]));
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const int gz = z0 + pl;                                    // wave-uniform
      const bool pok = gz >= 0 && gz < D;
#pragma unroll
      for (int it = 0; it < 3; ++it) {
        const bool ok = pok && pj_pix[it] >= 0 && !(WINO_PJ_ABL & 1);
        const int vox = gz * H * W + pj_pix[it];
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ra, ok ? vox * 64 + kg * 16 : 0x7fffffff, 0, 0);
        pj_yp[pl * 3 + it] = (f32x4){__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3])};
        const float nv = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rn, (ok && has_norm) ? vox * 4 : 0x7fffffff, 0, 0));
        pj_nr[pl * 3 + it] = (ok && has_norm) ? nv : 1.f;
      }
    }
  };
  auto pj_finish = [&](int half_base) {
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(pj_lo[i]));
#pragma unroll
   for (int pl0 = 0; pl0 < 2; ++pl0) {
    f32x4 g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr ((WINO_PJ_ABL & 2) == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k)
        g[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(pj_aw[pl0][i], pj_b[k][i], g[k], 0, 0, 0);
    } else {
#pragma unroll
      for (int k = 0; k < 3; ++k) g[k] = pj_b[k] + pj_aw[pl0];
    }
#pragma unroll
    for (int k = pl0 * 3; k < pl0 * 3 + 3; ++k) {
      const f32x4 yp = pj_yp[k];
      f32x4 v = g[k - pl0 * 3] * pj.he;
      if ((pj.flags & LF_EPI_PIXELNORM) && !(WINO_PJ_ABL & 2)) {
        float dot = v[0] * yp[0] + v[1] * yp[1] + v[2] * yp[2] + v[3] * yp[3];
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        dot *= (1.f / 16.f);
        const float rinv = fast_rcp(pj_nr[k]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (v[e] - yp[e] * dot) * rinv;
      }
      if (pj.flags & LF_EPI_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = yp[e] > 0.f ? v[e] : v[e] * slope;
      }
      const int pl = k / 3, it = k % 3;
      if (it < 2 || fa < 3 || pj_last_ok)
        *(f32x4*)(buf + half_base + pl * (HYw * HXw * 64) + (pj_lo[it] ^ (pl * 32))) = v;
    }
   }
  };
  // slide: the upper half's group regions of this wave -> the lower half, LDS -> LDS through three registers at a time (the
  // buffer is free once the tile's first barrier has passed, and the regions are this wave's own)
  auto pj_slide = [&]() {
    int sbase = fa * 1024 + lane * 16;
    asm volatile("" : "+v"(sbase));
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      u32x4 tmp[3];
#pragma unroll
      for (int it = 0; it < 3; ++it) tmp[it] = *(const u32x4*)(buf + HALFBw + pl * (HYw * HXw * 64) + it * 4096 + sbase);
#pragma unroll
      for (int it = 0; it < 3; ++it)
        if (it < 2 || fa < 3 || lane < 16) *(u32x4*)(buf + pl * (HYw * HXw * 64) + it * 4096 + sbase) = tmp[it];
    }
  };
  // FUSE = 1: per-wave projection accumulators (2 output rows x 16 x x 16 cout) and the A operands of the tile's two planes
  f32x4 pacc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};

  if constexpr (FUSE == 2) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int sl = (fa + 4 * it) * 16 + n;
      const int ly = sl / HXw;
      const int q0 = kg ^ ((((sl >> 2) & 1) << 1) | ((ly >> 1) & 1));
      pj_lo[it] = sl * 64 + q0 * 16;
    }
    pj_column(cx, cy, cn);
    pj_issue(cz * TZw - 1, cn);
    pj_issue_hot(cz * TZw - 1, cn);
    pj_finish(0);
    pj_issue(cz * TZw + 1, cn);
    pj_issue_hot(cz * TZw + 1, cn);
    pj_finish(HALFBw);
  } else {
    halo_fetch(cx, cy, cz, cn, true, false);
    halo_commit();
  }
  lds_barrier();

#if WINO_ABL & 16
#define TS(k) do { if (blockIdx.x == 11 && lane == 0) ((unsigned*)norm_out)[((t - t_begin) * 4 + fa) * 16 + (k)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define TS(k) do {} while (0)
#endif
  for (int t = t_begin; t < t_end; ++t) {
    TS(0);
#if WINO_ABL & 16
    unsigned* const tsp = (blockIdx.x == 11 && lane == 0) ? (unsigned*)norm_out + ((t - t_begin) * 4 + fa) * 16 : nullptr;
#else
    unsigned* const tsp = nullptr;
#endif
    switch (fa) {
      case 0: wino_compute<0, SPLIT, FUSE != 0>(buf, off, wt, whi, wlo, in_scale, px + fa * 8192, pw, tsp); break;
      case 1: wino_compute<1, SPLIT, FUSE != 0>(buf, off, wt, whi, wlo, in_scale, px + fa * 8192, pw, tsp); break;
      case 2: wino_compute<2, SPLIT, FUSE != 0>(buf, off, wt, whi, wlo, in_scale, px + fa * 8192, pw, tsp); break;
      default: wino_compute<3, SPLIT, FUSE != 0>(buf, off, wt, whi, wlo, in_scale, px + fa * 8192, pw, tsp); break;
    }
    TS(1);
    lds_barrier();                                  // every wave is done reading the halo; all partials are in LDS
    TS(2);
    __builtin_amdgcn_s_setprio(WINO_PN);

    // next tile's halo: in flight during the exchange / epilogue below and, for the CU, under the MFMA phase
    // of the other resident workgroup
    int nx = cx, ny = cy, nz = cz + 1, nn = cn;
    if (nz == tiles_z) { nz = 0; ++nx; }
    if (nx == tiles_x) { nx = 0; ++ny; }
    if (ny == tiles_y) { ny = 0; ++nn; }
    TS(3);

    // this wave finishes rows y = 2fa, 2fa+1 of the tile: lane = x*4 + channel quarter
    const int bx = cx, by = cy, bz = cz, tt = cn;
    const int gx = bx * TXw + ex;
    const bool xok = gx < W;
    int voxi[4];
    bool okv[4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int zo = 0; zo < 2; ++zo) {
        const int gz = bz * TZw + zo, gy = by * TYw + 2 * fa + j;       // wave-uniform: scalar unit
        const bool rowok = gy < H && gz < D;
        const int rowbase = rowok ? (gz * H + gy) * W : 0;
        okv[j * 2 + zo] = xok && rowok;
        voxi[j * 2 + zo] = okv[j * 2 + zo] ? rowbase + gx : 0;
      }
    f32x4 pyv[4];
    float pnv[4];
    if (prev_y != nullptr) {
      const float* pybase = prev_y + (long)tt * nvox * 16;
      const float* pnbase = prev_norm ? prev_norm + (long)tt * nvox : nullptr;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pyv[k] = okv[k] ? *(const f32x4*)(pybase + voxi[k] * 16 + eq * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        pnv[k] = (okv[k] && pnbase) ? pnbase[voxi[k]] : 1.f;
      }
    }
    // (bias is re-read per tile -- an L1 hit queued ahead of the halo -- rather than held in four registers)
    f32x4 bv4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const float* bp = bias;
      asm volatile("" : "+s"(bp));
      if (bp != nullptr) bv4 = *(const f32x4*)(bp + eq * 4);
    }
    TS(4);
    if constexpr (FUSE == 2) {
      // (the next halo is formed behind the epilogue, below)
    } else {
      halo_fetch(nx, ny, nz, nn, t + 1 < t_end, nz != 0);
    }
    TS(5);

    f32x4 o[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      f32x4 p[4];
      if constexpr ((WINO_ABL & 2) != 0) {
        o[j * 2 + 0] = *(const f32x4*)(px + fa * 8192 + fa * 2048 + j * 1024 + pr);
        o[j * 2 + 1] = *(const f32x4*)(px + fa * 8192 + ((fa + 1) & 3) * 2048 + j * 1024 + pr);
      } else {
#pragma unroll
        for (int a2 = 0; a2 < 4; ++a2)
          p[a2] = *(const f32x4*)(px + a2 * 8192 + fa * 2048 + j * 1024 + pr);
        o[j * 2 + 0] = p[0] + p[1] + p[2];                        // z output transform
        o[j * 2 + 1] = pk_sub(pk_sub(p[1], p[2]), p[3]);
      }
    }

    unsigned char* ybase = (unsigned char*)(y + (long)tt * nvox * 16) + eq * 16;
    float* nbase = norm_out ? norm_out + (long)tt * nvox : nullptr;
    f32x4 v[4];
    float rn[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rn[k] = 1.f;
      if constexpr ((WINO_ABL & 1) != 0) { v[k] = o[k]; continue; }
      if (prev_y == nullptr || addmode) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = o[k][e] * out_scale + bv4[e];
          if (addmode) u += pyv[k][e];
          if (flags & LF_EPI_LRELU) u = fmaxf(u, u * slope);
          v[k][e] = u;
          ss += u * u;
        }
        if (flags & LF_EPI_PIXELNORM) {
          const float tq = quad_sum(ss) * (1.f / 16.f) + eps;
          const float rinv = fast_rsqrt(tq);
          rn[k] = tq * rinv;
          v[k] *= rinv;
        }
      }
    }
    // the next halo (and, for the gradient form, the previous layer's activations) must have arrived
    // before the tile's last barrier; this tile's stores are issued after the wait so they stay out of it
    // and drain behind the next tile's MFMAs
    TS(6);
    if constexpr (FUSE != 2) halo_commit();                       // (FUSE = 2 forms its planes after the stores, below)
    TS(7);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (!(WINO_ABL & 1) && prev_y != nullptr && !addmode) {
        const f32x4 yp = pyv[k];
        v[k] = o[k] * out_scale;
        if (prev_flags & LF_EPI_PIXELNORM) {
          const float dot = quad_sum(v[k][0] * yp[0] + v[k][1] * yp[1] + v[k][2] * yp[2] + v[k][3] * yp[3]) * (1.f / 16.f);
          const float rinv = fast_rcp(pnv[k]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[k][e] = (v[k][e] - yp[e] * dot) * rinv;
        }
        if (prev_flags & LF_EPI_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[k][e] = yp[e] > 0.f ? v[k][e] : v[k][e] * slope;
        }
      }
      if (okv[k]) {
        __builtin_nontemporal_store(v[k], (f32x4*)(ybase + (unsigned)(voxi[k] * 64)));   // streamed: L2 is for halos
        if (!(WINO_ABL & 16) && (prev_y == nullptr || addmode) && (flags & LF_EPI_PIXELNORM) && nbase != nullptr && eq == 0) nbase[voxi[k]] = rn[k];
        if (amax_out != nullptr)
          wave_amax = fmaxf(wave_amax, fmaxf(fmaxf(fabsf(v[k][0]), fabsf(v[k][1])), fmaxf(fabsf(v[k][2]), fabsf(v[k][3]))));
      }
    }
    if constexpr (FUSE == 2) {
      // the next tile's two new planes, in a phase of their own behind this tile's stores: the kernel sits at the register
      // limit (254 of 256, no spills) and every placement that keeps the plane loads in flight under the exchange /
      // epilogue arithmetic spills 11 - 115 registers into the MFMA phase (measured, round 4); here nothing else is
      // live.  The loads' latency is exposed to this wave -- the SIMD's other wave (the CU's second workgroup) issues
      // its MFMAs meanwhile, which is what the two-workgroup organisation is for.
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < t_end && nz != 0) {
        if constexpr ((WINO_PJ_ABL & 4) == 0) pj_slide();
        pj_issue(nz * TZw + 1, nn);
        pj_issue_hot(nz * TZw + 1, nn);
        __builtin_amdgcn_sched_barrier(0);
        pj_finish(HALFBw);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FUSE == 1) {
      // finished records -> B-operand order through this wave's own (consumed) slices of the exchange region
      __builtin_amdgcn_sched_barrier(0);
      int lop = lane;
      asm volatile("" : "+v"(lop));                               // (opaque: or per-lane 64-bit addresses are hoisted out of the loop and spilled)
      const int trn = lop & 15, trk = lop >> 4;
      f32x4 pw_a[2];                                             // L2-resident weight slices of the tile's two planes
#pragma unroll
      for (int zo = 0; zo < 2; ++zo)
        pw_a[zo] = *(const f32x4*)((const char*)pj.wA + (long)min(bz * TZw + zo, D - 1) * 1024 + (unsigned)(lop * 16));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f32x4 vz = okv[k] ? v[k] : (f32x4){0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(px + (k >> 1) * 8192 + fa * 2048 + (k & 1) * 1024 + (lop >> 2) * 64 + (((lop & 3) ^ tr_swz(lop >> 2)) * 16)) = vz;
      }
      f32x4 bq[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        bq[k] = *(const f32x4*)(px + (k >> 1) * 8192 + fa * 2048 + (k & 1) * 1024 + trn * 64 + ((trk ^ tr_swz(trn)) * 16));
#pragma unroll
      for (int zo = 0; zo < 2; ++zo)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            pacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(pw_a[zo][i], bq[j * 2 + zo][i], pacc[j], 0, 0, 0);
      if (bz == tiles_z - 1) {                                   // top of the column: the projection's own epilogue
        f32x4 pb = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (pj.bias != nullptr) pb = *(const f32x4*)(pj.bias + trk * 4);
        const int pgx = bx * TXw + trn;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int pgy = by * TYw + 2 * fa + j;
          float ss = 0.f;
          f32x4 u;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float q = pacc[j][e] * pj.he + pb[e];
            if (pj.flags & LF_EPI_LRELU) q = lf_lrelu(q, slope);
            u[e] = q;
            ss += q * q;
          }
          float r = 1.f, rinv = 1.f;
          if (pj.flags & LF_EPI_PIXELNORM) {
            ss += __shfl_xor(ss, 16, 64);
            ss += __shfl_xor(ss, 32, 64);
            r = sqrtf(ss / 16.f + eps);
            rinv = 1.0f / r;
          }
          if (pgy < H && pgx < W) {
            const long pix = ((long)tt * H + pgy) * W + pgx;
            if (pj.flags & LF_EPI_PIXELNORM) { u[0] *= rinv; u[1] *= rinv; u[2] *= rinv; u[3] *= rinv; }
            *(f32x4*)(pj.zp + pix * 16 + trk * 4) = u;
            if ((pj.flags & LF_EPI_PIXELNORM) && pj.pnorm != nullptr && trk == 0) pj.pnorm[pix] = r;
          }
          pacc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    TS(8);
    lds_barrier();                                              // next halo visible to all; partials consumed
    cx = nx; cy = ny; cz = nz; cn = nn;
    if constexpr (FUSE == 2) {
      if (t + 1 < t_end && nz == 0) {                            // bottom of a new column: its pixels, all four halo planes
        pj_column(cx, cy, cn);
        pj_issue(-1, cn);
        pj_issue_hot(-1, cn);
        pj_finish(0);
        pj_issue(1, cn);
        pj_issue_hot(1, cn);
        pj_finish(HALFBw);
        lds_barrier();
      }
    }
  }
  if (amax_out != nullptr) {
    float m = wave_amax;
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) m = fmaxf(m, __shfl_xor(m, o2, 64));
    lf_amax_publish(amax_out, m, lane);
  }
}

__global__ void __launch_bounds__(256, 2) conv3d_c16_wino_kernel(
    const float* __restrict__ x, const float* __restrict__ upack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out, int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z,
    int ntiles, float he, unsigned flags, float slope, float eps, const float* __restrict__ prev_y,
    const float* __restrict__ prev_norm, unsigned prev_flags, float* __restrict__ amax_out) {
  conv3d_c16_wino_body<false>(x, upack, bias, y, norm_out, N, D, H, W, tiles_x, tiles_y, tiles_z, ntiles, he, flags, slope, eps,
                              prev_y, prev_norm, prev_flags, nullptr, amax_out);
}

__global__ void __launch_bounds__(256, 2) conv3d_c16_wino_f16x3_kernel(
    const float* __restrict__ x, const float* __restrict__ upack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out, int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z,
    int ntiles, float he, unsigned flags, float slope, float eps, const float* __restrict__ prev_y,
    const float* __restrict__ prev_norm, unsigned prev_flags, const float* __restrict__ amax_in,
    float* __restrict__ amax_out) {
  conv3d_c16_wino_body<true>(x, upack, bias, y, norm_out, N, D, H, W, tiles_x, tiles_y, tiles_z, ntiles, he, flags, slope, eps,
                             prev_y, prev_norm, prev_flags, amax_in, amax_out);
}

__global__ void __launch_bounds__(256, 2) conv3d_c16_wino_projfwd_kernel(
    const float* __restrict__ x, const float* __restrict__ upack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out, int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z,
    int ntiles, float he, unsigned flags, float slope, float eps, const float* __restrict__ prev_y,
    const float* __restrict__ prev_norm, unsigned prev_flags, WinoProj pj) {
  conv3d_c16_wino_body<false, 1>(x, upack, bias, y, norm_out, N, D, H, W, tiles_x, tiles_y, tiles_z, ntiles, he, flags, slope,
                                 eps, prev_y, prev_norm, prev_flags, nullptr, nullptr, pj);
}

__global__ void __launch_bounds__(256, 2) conv3d_c16_wino_projbwd_kernel(
    const float* __restrict__ x, const float* __restrict__ upack, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out, int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z,
    int ntiles, float he, unsigned flags, float slope, float eps, const float* __restrict__ prev_y,
    const float* __restrict__ prev_norm, unsigned prev_flags, WinoProj pj) {
  conv3d_c16_wino_body<false, 2>(x, upack, bias, y, norm_out, N, D, H, W, tiles_x, tiles_y, tiles_z, ntiles, he, flags, slope,
                                 eps, prev_y, prev_norm, prev_flags, nullptr, nullptr, pj);
}

template <bool COLUMNS = false, typename K, typename... Extra>
int launch_wino(K kernel, const float* x, const void* upack, const float* bias, float* y, float* norm_out, int N, int D,
                int H, int W, float he, unsigned flags, float slope, float eps, const float* prev_y, const float* prev_norm,
                unsigned prev_flags, void* stream, Extra... extra) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return LF_EINVAL;
  if ((long)D * H * W * 64 >= 0x7fffffffL || !(slope > 0.f && slope < 1.f)) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(y) || !lf_aligned16(upack) || (bias && !lf_aligned16(bias))) return LF_EALIGN;
  const bool add = prev_y != nullptr && (prev_flags & LF_EPI_ADD);
  if (add && prev_flags != LF_EPI_ADD) return LF_EINVAL;
  if (prev_y != nullptr && !add && (flags != 0 || bias != nullptr)) return LF_EINVAL;
  if ((prev_flags & LF_EPI_PIXELNORM) && prev_y != nullptr && prev_norm == nullptr) return LF_EINVAL;
  const int ptx = (W + TXw - 1) / TXw, pty = (H + TYw - 1) / TYw, ptz = (D + TZw - 1) / TZw;
  const long pt = (long)ptx * pty * ptz * N;
  if (pt > 0x7fffffffL) return LF_EINVAL;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const size_t shmem = (size_t)LDSw;
  static lf_devmask_t attr_set;                                   // (one instance per kernel: the template is per K)
  {
    hipError_t e = lf_ensure_dyn_lds(attr_set, (const void*)kernel, (int)shmem);
    if (e != hipSuccess) return (int)e;
  }
  const long want = 2L * cus;                                     // two resident workgroups per CU
  const long units = COLUMNS ? pt / ptz : pt;                     // (the fused forms hand out whole columns of tiles)
  const unsigned grid = (unsigned)(units < want ? units : want);
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), shmem, (hipStream_t)stream, x, (const float*)upack, bias, y, norm_out, N, D,
                     H, W, ptx, pty, ptz, (int)pt, he, flags, slope, eps, prev_y, prev_norm, prev_flags, extra...);
  return lf_launch_status();
}

}  // namespace

// floats of the transformed-weight pack: [4 z-freq][16 (y,x)-freq][4 k-chunks][64 lanes]
extern "C" size_t lf_conv3d_c16_wino_upack_floats(void) { return (size_t)4 * 16 * 4 * 64; }

extern "C" int lf_conv3d_c16_wino(const float* x, const float* upack, const float* bias, float* y, float* norm_out,
                                  int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                                  const float* prev_y, const float* prev_norm, unsigned prev_flags,
                                  float* amax_out, void* stream) {
  lf_clear_error();
  return launch_wino(conv3d_c16_wino_kernel, x, upack, bias, y, norm_out, N, D, H, W, he, flags, slope, eps, prev_y, prev_norm,
                     prev_flags, stream, amax_out);
}

// halfs of the split transformed-weight pack: [4 z-freq][16 (y,x)-freq][hi, lo][64 lanes][4 cin]
extern "C" size_t lf_conv3d_c16_wino_split_upack_halfs(void) { return (size_t)4 * 16 * 2 * 64 * 4; }

extern "C" int lf_conv3d_c16_wino_split(const float* x, const void* upack, const float* bias, float* y, float* norm_out,
                                        int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                                        const float* prev_y, const float* prev_norm, unsigned prev_flags,
                                        const float* amax_in, float* amax_out, void* stream) {
  lf_clear_error();
  return launch_wino(conv3d_c16_wino_f16x3_kernel, x, upack, bias, y, norm_out, N, D, H, W, he, flags, slope, eps, prev_y,
                     prev_norm, prev_flags, stream, amax_in, amax_out);
}

// floats of a projection slice pack of the fused forms: [D][64 lanes][4]
extern "C" size_t lf_conv3d_c16_wino_proj_pack_floats(int D) { return (size_t)(D > 0 ? D : 0) * 64 * 4; }

extern "C" int lf_conv3d_c16_wino_projfwd(const float* x, const float* upack, const float* bias, float* y, float* norm_out,
                                          int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                                          const float* proj_wA, const float* proj_bias, float* zp, float* pnorm,
                                          float proj_he, unsigned proj_flags, void* stream) {
  lf_clear_error();
  if (proj_wA == nullptr || zp == nullptr) return LF_EINVAL;
  if (!lf_aligned16(proj_wA) || !lf_aligned16(zp) || (proj_bias && !lf_aligned16(proj_bias))) return LF_EALIGN;
  if ((proj_flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) != 0) return LF_EINVAL;
  WinoProj pj = {proj_wA, proj_bias, zp, pnorm, nullptr, nullptr, proj_he, proj_flags};
  return launch_wino<true>(conv3d_c16_wino_projfwd_kernel, x, upack, bias, y, norm_out, N, D, H, W, he, flags, slope, eps,
                           nullptr, nullptr, 0u, stream, pj);
}

extern "C" int lf_conv3d_c16_wino_projbwd(const float* gp, const float* proj_wtA, float proj_he, const float* act,
                                          const float* act_norm, unsigned act_flags, const float* upack, float* y,
                                          int N, int D, int H, int W, float he, float slope, const float* prev_y,
                                          const float* prev_norm, unsigned prev_flags, void* stream) {
  lf_clear_error();
  if (gp == nullptr || proj_wtA == nullptr || act == nullptr) return LF_EINVAL;
  if ((act_flags & LF_EPI_PIXELNORM) && act_norm == nullptr) return LF_EINVAL;
  if ((act_flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) != 0) return LF_EINVAL;
  if (!lf_aligned16(gp) || !lf_aligned16(proj_wtA) || (prev_y && !lf_aligned16(prev_y))) return LF_EALIGN;
  if (prev_y != nullptr && (prev_flags & LF_EPI_ADD)) return LF_EINVAL;
  WinoProj pj = {proj_wtA, nullptr, nullptr, nullptr, gp, act_norm, proj_he, act_flags};
  return launch_wino<true>(conv3d_c16_wino_projbwd_kernel, act, upack, nullptr, y, nullptr, N, D, H, W, he, 0u, slope, 0.f,
                           prev_y, prev_norm, prev_flags, stream, pj);
}
