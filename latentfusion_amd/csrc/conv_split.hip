// Split-precision direct form of the dominant kernel: conv3d 16->16 on the f16 matrix cores with
// fp32-equivalent accuracy ("f16x3").
//
//   y = PixelNorm(LeakyReLU(conv3d(x, W) * he + b))            latentfusion/modules/blocks.py:152-158
//                                                              latentfusion/modules/equalized.py:57-64
// and, with transposed / flipped weights and `prev_*` set, the data gradient fused with the previous layer's
// LeakyReLU' / PixelNorm' (autograd of the same lines).
//
// Every fp32 operand x is split as x = hi + lo with hi = (f16)x, lo = (f16)(x - hi): 22 mantissa bits.
// The product is formed from three f16 MFMAs accumulating in fp32,
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (dropped a_lo*b_lo <= 2^-22 |a b|),
// each partial product being exact in fp32 (11 x 11 significant bits).  The f16 MFMA runs at 16x the fp32 MFMA
// rate and on its own pipe (the fp32 MFMA shares the fp32 VALU's), so the convolution becomes a balanced
// HBM / LDS / matrix-pipe problem instead of an issue-bound one.  Activations stay fp32 in HBM: the split happens
// on-chip while the halo is staged.
//
// Range: forward activations are O(1) (PixelNorm output) and need no scaling.  Gradients can be tiny, so
// data-gradient launches multiply the input by a power of two derived from the tensor's max-abs (`amax_in`,
// produced by the previous kernel with an order-independent atomic max) and undo it exactly in the epilogue;
// elements far below the tensor max lose relative but not absolute accuracy, which is what a dot product needs.
//
// Organisation (gfx950):
//   * 256-thread workgroups, TWO per CU (2 x 69,696 B of LDS), each walking up columns of 2 x 8 x 16-voxel tiles;
//     wave w owns z plane w & 1, rows 4 (w >> 1) .. +3 of the tile: 4 rows x 16 voxels x 16 couts = 4 accumulators.
//   * the halo lives in LDS as f16 hi / lo planes (32 B per voxel and plane kind, conflict-free for the
//     ds_read_b128 operand reads without a swizzle) in a RING of 6 z-plane slots: a tile reads 4 planes (4 x 10 x 18
//     voxels), the upper two of which are the lower two of the next tile up the column, and the two planes the next
//     tile adds (23 KB of fp32) go to the two slots NOT being read: they are requested before the tile's MFMA phase
//     (6 x 1 KiB buffer loads per wave), converted and written behind it, and one workgroup barrier per tile
//     separates "everyone has read planes 0, 1" / "the new planes are visible" from the next tile.
//   * taps are paired into K = 32 MFMAs (lane groups kg 0,1: first tap, 2,3: second tap) so that every B operand
//     is a plain 16-lane-contiguous read, and so that ONE operand serves up to three output rows:
//       pairs 0-8  : (kz=0,ky,kx) + (kz=1,ky,kx)      operand P[h][kx] = halo row h of planes p / p+1, used by rows
//                                                      h, h-1, h-2 with the weights of ky = 0, 1, 2
//       pairs 9-11 : (kz=2,ky,0) + (kz=2,ky,1)        operand Q[h]     = halo row h of plane p+2 at x, x+1
//       pair 12    : (kz=2,0,2) + (kz=2,1,2)          operand R[h]     = plane p+2, rows h / h+1 at x+2
//       pair 13    : (kz=2,2,2) + zero weights        operand R[h+2]
//     = 30 operand reads (hi + lo: 60 ds_read_b128) for the 168 MFMAs of a wave and tile (the previous version of
//     this kernel: 112 reads), weights (14 pairs x hi / lo) resident in 112 VGPRs.  The reads run two operands ahead of
//     their MFMAs, pinned with sched_barriers (the scheduler otherwise sinks every read to its first use).
//   * a wave alone issues dependent non-MFMA work at ~10 cycles per instruction and its SIMD partner (the other
//     workgroup) does not speed that up, so the per-tile instruction stream outside the MFMAs is what is left to cut:
//     the finished tile's epilogue (scale, bias, LeakyReLU, PixelNorm via v_permlane swaps, 1 KiB buffer stores) rides
//     between the NEXT tile's MFMAs, and all addressing is per COLUMN of tiles (per-lane offsets, out-of-volume lanes
//     = out-of-range offsets that the buffer hardware drops) with the z plane as the instructions' scalar offset.
//   Measured (tools/split_ab.py, tools/split_timeline.py, 8 x 128^3): 0.65-0.75 ms per launch (previous version 1.0-1.1),
//   MFMA phase 2.9k of 7.8k cycles per tile and wave.  tools/ub/coexec.hip: plain fp32 VALU work of the partner wave is
//   free beside f16 MFMAs, packed-fp32 (v_pk_*) and v_fma_mix work is not.
//   This file is compiled with -fno-slp-vectorize (build.py): SLP-packed fp32 arithmetic inside the MFMA phase gave rare
//   run-to-run differences in the gradient form (tests/test_engine_gpu.py::test_split_conv3d_is_run_to_run_identical).
#include <type_traits>
#include "lf_common.h"
#include "ring_tile.h"


#ifndef SPLIT_PF
#define SPLIT_PF 2                                       // operand reads issued this many operands ahead of their MFMAs
#endif
#ifndef SPLIT_ABL
#define SPLIT_ABL 0                                      // ablations (tools/split_ab.py; 64 = every MFMA twice): 1 no MFMAs, 2 no operand reads, 4 no halo fetch, 8 no stores, 16 no commit; 32 = cycle stamps (tools/split_timeline.py)
#endif
#ifndef SPLIT_E0
#define SPLIT_E0 1                                       // operand step at which the previous tile's epilogue starts (8 steps)
#endif
#ifndef SPLIT_C0
#define SPLIT_C0 100                                     // operand step at which the conversion of the incoming planes starts (6 steps); >= 32: behind the MFMA phase (faster)
#endif
#ifndef SPLIT_WGS
#define SPLIT_WGS 2                                      // resident workgroups per CU (1: timeline without a SIMD partner)
#endif

namespace {

// x -> f16 hi = f16(x s), lo = f16(x s - hi) for the four floats of a staged piece.  Written as fp32 FMAs on purpose: with
// SLP vectorisation off (build.py compiles this file with -fno-slp-vectorize) they become the mixed-precision FMAs
// v_fma_mix{lo,hi}_f16, which read an f16 half as a source and write the f16 result into one half of the destination
// (10 VALU instructions per piece; the convert / convert back / subtract / convert form costs 12 and packed-f32 ops).
// NOT inline asm: the hazard recogniser does not look inside asm statements, and a VALU write from one into a register
// that an in-flight MFMA still reads as its accumulator input (the allocator recycles those freely inside the MFMA
// phase) corrupted the last rows of that MFMA's result now and then.
typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split_piece(const f32x4 v, float s, f16x4s& hi, f16x4s& lo) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    hi[c] = (_Float16)__builtin_fmaf(v[c], s, 0.f);
    lo[c] = (_Float16)__builtin_fmaf(v[c], s, -(float)hi[c]);
  }
}

// NP = 3: the f16 hi / lo split above.  NP = 1: the bf16-autocast policy of the training step on the same ring -- one bf16
// piece per operand (RNE while staging: what autocast's cast does), one v_mfma_f32_16x16x32_bf16 per product, no lo planes
// (35 KB of LDS), no pre-scaling (bf16 has fp32's range); `round_out` = 1 rounds the result the way autocast's
// half-precision convolution and `* he` do (conv_bf16.hip); GRAD with LF_EPI_ADD in prev_flags adds prev_y instead of
// applying the previous layer's backward epilogue (the addend form of the ConvGRU gates).
// IO (NP = 1 only; round 5): storage type of the three volumes -- bit 0: the input x, bit 1: the output y, bit 2: prev_y
// (the addend) are bf16 channels-last records of 32 B per voxel instead of fp32 records of 64 B.  An input that is only
// ever read through this staging (which rounds to bf16) loses nothing by being stored rounded; half the bytes move.
template <bool GRAD, int NP, int IO = 0>
__global__ void __launch_bounds__(256, 2) conv3d_c16_f16x3_kernel(
    const float* __restrict__ x, const void* __restrict__ wsplit_v, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z, int ntiles,
    float he, unsigned flags, float slope, float eps,
    const float* __restrict__ prev_y, const float* __restrict__ prev_norm, unsigned prev_flags,
    const float* __restrict__ amax_in, float* __restrict__ amax_out, int round_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using h8 = std::conditional_t<NP == 1, bf16x8s, f16x8>;
  using h1 = std::conditional_t<NP == 1, __bf16, _Float16>;
  constexpr int LDS_B = NP == 1 ? LO_OFF + GUARD_B : LDSs;
  static_assert(IO == 0 || NP == 1, "bf16 storage is a property of the bf16 form");
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0, PREV16 = (IO & 4) != 0;
  constexpr int IN_SH = IN16 ? 5 : 6;                     // log2 bytes per input voxel record
  const h1* wsplit = (const h1*)wsplit_v;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pz = wv & 1, ry = wv >> 1;                    // this wave's output plane and row quad; also its fetch share
  const int n = lane & 15, kg = lane >> 4;
  // zero-weight K slots and the rows an operand reads past its wave's share meet whatever is in LDS: 0 * garbage
  // could be NaN, so everything starts as zeros (the planes hold finite f16 values from then on)
  for (int i = tid; i < LDS_B / 16; i += 256) ((u32x4s*)smem)[i] = (u32x4s){0u, 0u, 0u, 0u};
  __syncthreads();

  // workgroups b, b+8, b+16, ... run on the same XCD (one L2 each): give them consecutive tile ranges so that the
  // halo columns shared by neighbouring ranges are fetched from HBM once
  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;
  const int per = (ntiles + nb - 1) / nb;
  const int t_begin = lb * per;
  const int t_end = min(t_begin + per, ntiles);
  if (t_begin >= t_end) return;

  const long nvox = (long)D * H * W;
  const unsigned sample_bytes = (unsigned)(nvox * 64);

  // power-of-two input scale from the tensor's max-abs (gradient launches); 1 otherwise
  float in_scale = 1.f;
  if (NP != 1 && amax_in != nullptr) {
    const float am = lf_amax_read(amax_in, lane);
    if (am > 0.f && am < 3.0e38f) {
      int ex;
      frexpf(am, &ex);                                               // am = m * 2^ex, m in [0.5, 1)
      in_scale = ldexpf(1.f, 13 - ex);                               // max maps into [2^12, 2^13)
    }
  }
  const float out_scale = he / in_scale;
  float scale_s = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, in_scale)));
  asm volatile("" : "+s"(scale_s));                               // (opaque: an FMA by a visible 1.0 would be folded into converts)

  // ---- weights (hi, lo) -> registers: [pair][term][cout 16][k 32] halfs; A operand: lane (cout n, k = kg*8..+7) ----
  h8 whi[NPAIR], wlo[NP == 1 ? 1 : NPAIR];
  {
    const h1* wl = wsplit + n * 32 + kg * 8;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      if constexpr (NP == 1) {
        whi[p] = *(const h8*)(wl + p * 512);
      } else {
        whi[p] = *(const h8*)(wl + (p * 2 + 0) * 512);
        wlo[p] = *(const h8*)(wl + (p * 2 + 1) * 512);
      }
    }
  }

  // ---- halo pieces: wave w stages rows 5 ry .. +4 of incoming plane pz.  Piece it < 5 = row 5 ry + it, x = lane >> 2
  // (0..15), quarter lane & 3: 1 KiB contiguous in HBM, 512 B contiguous in each LDS plane kind; piece 5 = x 16, 17
  // of the five rows (lanes 0..39: row 5 ry + (lane >> 3)).
  // Addressing is split into what changes per COLUMN of tiles and what changes per tile: foff[] = per-lane byte offsets
  // inside one z plane of the sample (0x80000000 where the halo leaves the volume in x or y: the buffer load then
  // returns zeros), recomputed when the walk enters a new column; the plane itself is the instruction's SCALAR offset,
  // and a plane outside [0, D) -- wave-uniform, every wave fetches one plane -- gets a zero-sized descriptor. ----
  constexpr int OOB = (int)0x80000000;
  const int frow0 = 5 * ry;
  const int erow = frow0 + (lane >> 3), ecol = 16 + ((lane >> 2) & 1);
  const int eldso = (erow * HXs + ecol) * 32 + (lane & 3) * 8;
  const bool e_ok = lane < 40;
  const int plane_bytes = H * W * 64;                     // (fp32 records: the output side; halved per use for bf16 storage)
  const int in_plane_bytes = (H * W) << IN_SH;
  const unsigned in_sample_bytes = (unsigned)(nvox << IN_SH);

  int foff[NPIECE];
  const float* f_x = x;
  auto fetch_column = [&](int bx, int by, int bn) {
    const int ox = bx * TXs - 1, oy = by * TYs - 1;
    const int col = ox + (lane >> 2);
#pragma unroll
    for (int it = 0; it < 5; ++it) {
      const int row = oy + frow0 + it;
      foff[it] = ((unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H) ? ((row * W + col) << IN_SH) + ((lane & 3) << (IN_SH - 2)) : OOB;
    }
    const int row = oy + erow, c2 = ox + ecol;
    foff[5] = (e_ok && (unsigned)c2 < (unsigned)W && (unsigned)row < (unsigned)H) ? ((row * W + c2) << IN_SH) + ((lane & 3) << (IN_SH - 2)) : OOB;
    f_x = x + (long)bn * nvox * (IN16 ? 8 : 16);          // (x is declared float*: a bf16 record is 8 floats' worth of bytes)
  };
  u32x4s stg[NPIECE];
  auto fetch_plane = [&](int z, bool on) {
    const bool v = on && (unsigned)z < (unsigned)D;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)f_x, 0, v ? in_sample_bytes : 0u, 0x00020000);
    const int soff = v ? z * in_plane_bytes : 0;
#pragma unroll
    for (int it = 0; it < NPIECE; ++it) {
      if constexpr (IN16) {
        const u32x2s h = __builtin_amdgcn_raw_buffer_load_b64(rs, foff[it], soff, 0);
        stg[it] = (u32x4s){h[0], h[1], 0u, 0u};
      } else {
        stg[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, foff[it], soff, 0);
      }
    }
  };
  // fp32 -> f16 hi / lo of staged piece `it`, into the plane slot at dst
  auto commit_piece = [&](unsigned char* dst, auto itc) {
    constexpr int it = decltype(itc)::v;
    if constexpr (NP == 1) {
      bf16x4s h;
      if constexpr (IN16) h = __builtin_bit_cast(bf16x4s, (u32x2s){stg[it][0], stg[it][1]});
      else h = __builtin_convertvector(__builtin_bit_cast(f32x4, stg[it]), bf16x4s);
      if constexpr (it < 5) *(bf16x4s*)(dst + (frow0 + it) * (HXs * 32) + lane * 8) = h;
      else if (e_ok) *(bf16x4s*)(dst + eldso) = h;
    } else {
      f16x4s h, l;
      split_piece(__builtin_bit_cast(f32x4, stg[it]), scale_s, h, l);
      if constexpr (it < 5) {
        *(f16x4s*)(dst + (frow0 + it) * (HXs * 32) + lane * 8) = h;
        *(f16x4s*)(dst + (frow0 + it) * (HXs * 32) + lane * 8 + LO_OFF) = l;
      } else if (e_ok) {
        *(f16x4s*)(dst + eldso) = h;
        *(f16x4s*)(dst + eldso + LO_OFF) = l;
      }
    }
  };
  auto commit_plane = [&](int slot) {
    unsigned char* const dst = smem + slot * PLANE_B;
    static_for<0, NPIECE>([&](auto itc) { commit_piece(dst, itc); });
  };

  // tile coordinates are stepped, not divided: (cx, cy, cz, cn) = tile t, (nx, ny, nz, nn) = tile t + 1
  int cx, cy, cz, cn;
  {
    int tt = t_begin;                                            // z fastest: a workgroup walks up columns of tiles
    cz = tt % tiles_z; tt /= tiles_z;
    cx = tt % tiles_x; tt /= tiles_x;
    cy = tt % tiles_y; cn = tt / tiles_y;
  }
  // ring state: halo plane hp (0..3) of the current tile sits in slot (rot + hp) % 6
  int rot = 0;
  fetch_column(cx, cy, cn);
  fetch_plane(cz * TZs - 1 + pz, true);
  commit_plane(pz);
  fetch_plane(cz * TZs + 1 + pz, true);
  commit_plane(2 + pz);
  lds_barrier_s();

  f32x4 bv4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (!GRAD && bias != nullptr) bv4 = *(const f32x4*)(bias + kg * 4);
  const int lane_b = (4 * ry * HXs + n) * 32 + (kg & 1) * 16;     // B operand: halo row 4 ry (+ h), voxel n, channel half kg & 1
  float wave_amax = 0.f;

  // ---- epilogue of one tile, in 2 parts per output row so that it can be spread over the NEXT tile's MFMA phase (a
  // wave alone issues dependent VALU work at ~10 cycles per instruction; between MFMAs it is nearly free).  Lane holds
  // couts kg*4..+3 of voxel (gz, gy0 + r, gx).  Same addressing split as the halo: eoff[] = per-lane byte offset of
  // (gy0 + r, gx, couts kg*4..) inside a z plane of the sample, per column (OOB outside the volume: the store is
  // dropped, loads return 0); plane gz through the scalar offset; gz >= D: zero-sized descriptors. ----
  int eoff[RYs];
#pragma unroll
  for (int r = 0; r < RYs; ++r) eoff[r] = OOB;
  const int kgmask = kg == 0 ? 0 : OOB;                           // one lane per voxel writes the norm
  const float *e_y = y, *e_n = norm_out, *e_py = prev_y, *e_pn = prev_norm;
  auto epi_column = [&](int bx, int by, int bn) {
    const int gx = bx * TXs + n, gy0 = by * TYs + RYs * ry;
#pragma unroll
    for (int r = 0; r < RYs; ++r) eoff[r] = (gx < W && gy0 + r < H) ? ((gy0 + r) * W + gx) * 64 + kg * 16 : OOB;
    e_y = y + (long)bn * nvox * (OUT16 ? 8 : 16);
    e_n = norm_out ? norm_out + (long)bn * nvox : nullptr;
    if constexpr (GRAD) {
      e_py = prev_y + (long)bn * nvox * (PREV16 ? 8 : 16);
      e_pn = prev_norm ? prev_norm + (long)bn * nvox : nullptr;
    }
  };
  struct Epi { __amdgpu_buffer_rsrc_t rs_y, rs_n; int soff, soff_n; bool zv; };
  f32x4 ev[RYs];                                                  // part 0 -> part 1
  float et[RYs];
  f32x4 pyv[RYs];
  float pnv[RYs];
  auto epi_tile = [&](Epi& E, int bz, bool valid) {
    const int gz = bz * TZs + pz;
    E.zv = valid && gz < D;
    E.soff = E.zv ? gz * plane_bytes : 0;
    E.soff_n = E.zv ? gz * (plane_bytes >> 4) : 0;
    E.rs_y = __builtin_amdgcn_make_buffer_rsrc((void*)e_y, 0, E.zv ? (OUT16 ? sample_bytes >> 1 : sample_bytes) : 0u, 0x00020000);
    E.rs_n = __builtin_amdgcn_make_buffer_rsrc((void*)e_n, 0, (E.zv && e_n != nullptr) ? (sample_bytes >> 4) : 0u, 0x00020000);
    if constexpr (GRAD) {
      const __amdgpu_buffer_rsrc_t rs_py = __builtin_amdgcn_make_buffer_rsrc((void*)e_py, 0, E.zv ? (PREV16 ? sample_bytes >> 1 : sample_bytes) : 0u, 0x00020000);
      const __amdgpu_buffer_rsrc_t rs_pn = __builtin_amdgcn_make_buffer_rsrc((void*)e_pn, 0, (E.zv && e_pn != nullptr) ? (sample_bytes >> 4) : 0u, 0x00020000);
#pragma unroll
      for (int r = 0; r < RYs; ++r) {
        if constexpr (PREV16) {
          // (an out-of-volume offset 0x80000000 shifts to 0xC0000000: still past every buffer)
          const bf16x4s pb = __builtin_bit_cast(bf16x4s, __builtin_amdgcn_raw_buffer_load_b64(rs_py, eoff[r] >> 1, E.soff >> 1, 0));
          pyv[r] = __builtin_convertvector(pb, f32x4);
        } else {
          pyv[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_py, eoff[r], E.soff, 0));
        }
        pnv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_pn, (eoff[r] >> 4) & ~3, E.soff_n, 0));
      }
    }
  };
  auto epi_part = [&](const Epi& E, const f32x4 (&a)[RYs], auto rc, auto pc) {
    constexpr int r = decltype(rc)::v, part = decltype(pc)::v;
    if constexpr (part == 0) {
      if constexpr (GRAD) {
        ev[r] = a[r] * out_scale;
        et[r] = quarter_sum(ev[r][0] * pyv[r][0] + ev[r][1] * pyv[r][1] + ev[r][2] * pyv[r][2] + ev[r][3] * pyv[r][3]) * (1.f / 16.f);
      } else {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = a[r][e] * out_scale;
          if (NP == 1 && round_out) u = rb16(rb16(a[r][e]) * out_scale);   // autocast: half conv result, `* he` in half, fp32 bias
          u += bv4[e];
          if (flags & LF_EPI_LRELU) u = fmaxf(u, u * slope);
          ev[r][e] = u;
          ss += u * u;
        }
        et[r] = quarter_sum(ss) * (1.f / 16.f) + eps;
      }
    } else {
      f32x4 v = ev[r];
      float rn = 1.f;
      if constexpr (GRAD) {
        const f32x4 yp = pyv[r];
        if (prev_flags & LF_EPI_ADD) v += yp;
        if (prev_flags & LF_EPI_PIXELNORM) {
          const float rinv = fast_rcp_s(pnv[r]);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (v[e] - yp[e] * et[r]) * rinv;
        }
        if (prev_flags & LF_EPI_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = yp[e] > 0.f ? v[e] : v[e] * slope;
        }
      } else if (flags & LF_EPI_PIXELNORM) {
        const float rinv = fast_rsqrt_s(et[r]);
        rn = et[r] * rinv;
        v *= rinv;
      }
      if (!(SPLIT_ABL & 8) || v[0] == 123.456f) {
        if constexpr (OUT16)
          __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, __builtin_convertvector(v, bf16x4s)), E.rs_y, eoff[r] >> 1, E.soff >> 1, 2);
        else
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), E.rs_y, eoff[r], E.soff, 2);   // nt: L2 is for halos
        if (!(SPLIT_ABL & 32) && !GRAD && (flags & LF_EPI_PIXELNORM))
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rn), E.rs_n, (eoff[r] >> 4) | kgmask, E.soff_n, 0);
      }
      if (amax_out != nullptr) {
        const float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
        wave_amax = fmaxf(wave_amax, (E.zv && eoff[r] >= 0) ? m : 0.f);      // (lanes outside the volume hold junk)
      }
    }
  };

#if SPLIT_ABL & 32
#define TS(k) do { if (blockIdx.x == 11 && lane == 0) ((unsigned*)norm_out)[((t - t_begin) * 4 + wv) * 8 + (k)] = (unsigned)__builtin_readcyclecounter(); } while (0)
#else
#define TS(k) do {} while (0)
#endif
  f32x4 accP[RYs];                                                // the previous tile's sums, finished under this tile's MFMAs
#pragma unroll
  for (int r = 0; r < RYs; ++r) accP[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int px_ = cx, py_ = cy, pz_ = cz, pn_ = cn;
  for (int t = t_begin; t < t_end; ++t) {
    TS(0);
    int nx = cx, ny = cy, nz = cz + 1, nn = cn;
    if (nz == tiles_z) { nz = 0; ++nx; }
    if (nx == tiles_x) { nx = 0; ++ny; }
    if (ny == tiles_y) { ny = 0; ++nn; }
    const bool on = t + 1 < t_end;
    const bool slide = on && nz != 0;
    // per-column addressing, recomputed when the previous tile (epilogue) / the next tile (halo) starts a column
    if (t == t_begin + 1 || (t > t_begin && pz_ == 0)) epi_column(px_, py_, pn_);
    if (on && nz == 0) fetch_column(nx, ny, nn);
    Epi E;
    epi_tile(E, pz_, t > t_begin);                     // (gradient form: the previous layer's activations, requested first)
    // the two planes the next tile adds (slide: its halo planes 2, 3; new column: its planes 0, 1): in flight under
    // the MFMA phase, bound for the two ring slots this tile does not read
    if (!(SPLIT_ABL & 4)) fetch_plane(nz * TZs - 1 + (slide ? 2 : 0) + pz, on);
    __builtin_amdgcn_sched_barrier(0);                 // (the scheduler would sink the loads to their use, behind the MFMAs)
    TS(1);

    // ---- operand base addresses of this wave: planes pz, pz+1 (pairs along z) and pz+2 ----
    const int s0 = mod6(rot + pz), s1 = mod6(s0 + 1), s2 = mod6(s1 + 1);
    const int aP = ((kg >> 1) ? s1 : s0) * PLANE_B + lane_b;
    const int aQ = s2 * PLANE_B + lane_b + (kg >> 1) * 32;
    const int aR = s2 * PLANE_B + lane_b + 2 * 32 + (kg >> 1) * (HXs * 32);
    unsigned char* const cdst = smem + mod6(rot + 4 + pz) * PLANE_B;

    f32x4 acc[RYs];
#pragma unroll
    for (int r = 0; r < RYs; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      h8 bh[SPLIT_PF + 1], bl[NP == 1 ? 1 : SPLIT_PF + 1];
      static_for<0, NOP + SPLIT_PF>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        if constexpr (i < NOP && !(SPLIT_ABL & 2)) {
          constexpr int cls = op_cls(i), h = op_h(i), kx = op_kx(i);
          const int addr = cls == 0 ? aP : (cls == 1 ? aQ : aR);
          constexpr int imm = (h * HXs + kx) * 32;
          bh[i % (SPLIT_PF + 1)] = *(const h8*)(smem + addr + imm);
          if constexpr (NP != 1) bl[i % (SPLIT_PF + 1)] = *(const h8*)(smem + addr + imm + LO_OFF);
        }
        if constexpr (i >= SPLIT_PF && !(SPLIT_ABL & 1)) {
          constexpr int j = i - SPLIT_PF;
          constexpr int cls = op_cls(j), h = op_h(j), kx = op_kx(j);
          const h8 vh = bh[j % (SPLIT_PF + 1)], vl = bl[NP == 1 ? 0 : j % (SPLIT_PF + 1)];
          // up to three (row, pair) uses of this operand; the three product terms are issued term-major so that
          // consecutive MFMAs go to different accumulators
          static_for<0, 9>([&](auto uc) {
            constexpr int term = decltype(uc)::v / 3, u = decltype(uc)::v % 3;
            constexpr int r = cls < 2 ? h - u : (u == 0 ? h : (u == 1 ? h - 2 : -1));
            constexpr int p = cls == 0 ? kx * 3 + u : (cls == 1 ? 9 + u : 12 + u);
            if constexpr (r >= 0 && r < RYs) {
              if constexpr (term == 0 && NP != 1) acc[r] = mfma_k32(whi[p], vl, acc[r]);
              if constexpr (term == 1 && NP != 1) acc[r] = mfma_k32(wlo[NP == 1 ? 0 : p], vh, acc[r]);
              if constexpr (term == 2) acc[r] = mfma_k32(whi[p], vh, acc[r]);
              if constexpr ((SPLIT_ABL & 64) != 0) {         // every product twice: the MFMA count of a bf16 x 3 (six-product) split
                if constexpr (term == 0 && NP != 1) acc[r] = mfma_k32(whi[p], vl, acc[r]);
                if constexpr (term == 1 && NP != 1) acc[r] = mfma_k32(wlo[NP == 1 ? 0 : p], vh, acc[r]);
                if constexpr (term == 2) acc[r] = mfma_k32(whi[p], vh, acc[r]);
              }
            }
          });
        }
        // the previous tile's epilogue rides in the issue slots between the MFMAs (its inputs are ready)
        if constexpr (i >= SPLIT_E0 && i < SPLIT_E0 + 2 * RYs) epi_part(E, accP, IC<(i - SPLIT_E0) / 2>{}, IC<(i - SPLIT_E0) % 2>{});
        if constexpr (i >= SPLIT_C0 && i < SPLIT_C0 + NPIECE && !(SPLIT_ABL & 16)) commit_piece(cdst, IC<i - SPLIT_C0>{});
        // keep the software pipeline as written: left alone, the scheduler moves every read next to its first use
        // and the wave then waits out the full LDS latency once per operand
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    TS(2);
    if constexpr (SPLIT_E0 >= NOP + SPLIT_PF)          // (A/B: epilogue / conversion behind the MFMA phase instead of inside it)
      static_for<0, 2 * RYs>([&](auto ic) { epi_part(E, accP, IC<decltype(ic)::v / 2>{}, IC<decltype(ic)::v % 2>{}); });
#if SPLIT_ABL & 32
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TS(3);
#endif
    // the next tile's new planes: converted and written to the two free ring slots (last tile: zeros nobody reads)
    if constexpr (SPLIT_C0 >= NOP + SPLIT_PF && !(SPLIT_ABL & 16)) static_for<0, NPIECE>([&](auto itc) { commit_piece(cdst, itc); });
    TS(4);
    lds_barrier_s();                                   // this tile's planes 0, 1 are free; the new planes are visible
    TS(6);
    if (on && !slide) {
      // bottom of a new column: what was fetched are its planes 0, 1 (now in slots rot+4, rot+5); planes 2, 3 go to
      // the slots this tile has just released (exposed once per column)
      rot = mod6(rot + 4);
      fetch_plane(nz * TZs + 1 + pz, true);
      commit_plane(mod6(rot + 2 + pz));
      lds_barrier_s();
    } else {
      rot = mod6(rot + 2);
    }
#pragma unroll
    for (int r = 0; r < RYs; ++r) accP[r] = acc[r];
    px_ = cx; py_ = cy; pz_ = cz; pn_ = cn;
    cx = nx; cy = ny; cz = nz; cn = nn;
  }
  {                                                    // the last tile's epilogue
    if (t_end - t_begin == 1 || pz_ == 0) epi_column(px_, py_, pn_);
    Epi E;
    epi_tile(E, pz_, true);
    static_for<0, 2 * RYs>([&](auto ic) { epi_part(E, accP, IC<decltype(ic)::v / 2>{}, IC<decltype(ic)::v % 2>{}); });
  }
  if (amax_out != nullptr) {
    float m = wave_amax;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    lf_amax_publish(amax_out, m, lane);
  }
}

}  // namespace

extern "C" size_t lf_conv3d_c16_split_wpack_halfs(void) { return (size_t)NPAIR * 2 * 16 * 32; }

// taps[2*p], taps[2*p+1] = tap indices (kz*9 + ky*3 + kx) feeding K-slots [0,16) and [16,32) of pair p; -1 = zero weights
extern "C" void lf_conv3d_c16_split_pairs(int* taps) {
  for (int p = 0; p < NPAIR; ++p) { taps[2 * p] = pair_first(p); taps[2 * p + 1] = pair_second(p); }
}

extern "C" int lf_conv3d_c16_split(const float* x, const void* wsplit, const float* bias, float* y, float* norm_out,
                                   int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                                   const float* prev_y, const float* prev_norm, unsigned prev_flags,
                                   const float* amax_in, float* amax_out, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return LF_EINVAL;
  if ((long)D * H * W * 64 >= 0x7fffffffL || !(slope > 0.f && slope < 1.f)) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(y) || !lf_aligned16(wsplit) || (bias && !lf_aligned16(bias))) return LF_EALIGN;
  if (prev_y != nullptr && (flags != 0 || bias != nullptr)) return LF_EINVAL;
  if ((prev_flags & LF_EPI_PIXELNORM) && prev_y != nullptr && prev_norm == nullptr) return LF_EINVAL;
  const int ptx = (W + TXs - 1) / TXs, pty = (H + TYs - 1) / TYs, ptz = (D + TZs - 1) / TZs;
  const long pt = (long)ptx * pty * ptz * N;
  if (pt > 0x7fffffffL) return LF_EINVAL;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const size_t shmem = (size_t)LDSs;
  typedef void (*kern_t)(const float*, const void*, const float*, float*, float*, int, int, int, int, int, int, int, int, float,
                         unsigned, float, float, const float*, const float*, unsigned, const float*, float*, int);
  static const kern_t kerns[2] = {conv3d_c16_f16x3_kernel<false, 3>, conv3d_c16_f16x3_kernel<true, 3>};
  static lf_devmask_t attr_set[2];
  for (int i = 0; i < 2; ++i) {
    hipError_t e = lf_ensure_dyn_lds(attr_set[i], (const void*)kerns[i], (int)shmem);
    if (e != hipSuccess) return (int)e;
  }
  const long want = (long)SPLIT_WGS * cus;                        // two resident workgroups per CU
  const unsigned grid = (unsigned)(pt < want ? pt : want);
  const kern_t kern = kerns[prev_y != nullptr ? 1 : 0];
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, (hipStream_t)stream, x, wsplit, bias, y, norm_out, N, D, H,
                     W, ptx, pty, ptz, (int)pt, he, flags, slope, eps, prev_y, prev_norm, prev_flags, amax_in, amax_out, 0);
  return lf_launch_status();
}

static int g_ring_bf16_wgs = 2;
int lf_internal_ring_bf16_set_wgs(int v) {
  const int prev = g_ring_bf16_wgs;
  if (v == 2 || v == 3) g_ring_bf16_wgs = v;
  return prev;
}

// bf16 elements of the weight pack of the bf16 form: [14 pairs][16 cout][32 = 2 taps x 16 cin] (pairs: lf_conv3d_c16_split_pairs)
extern "C" size_t lf_conv3d_c16_ring_bf16_wpack_elems(void) { return (size_t)NPAIR * 16 * 32; }

static int ring_bf16_launch(const void* x, const void* wpack, const float* bias, void* y, float* norm_out,
                            int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                            const void* addend, int round_out, int io, void* stream);

extern "C" int lf_conv3d_c16_ring_bf16(const float* x, const void* wpack, const float* bias, float* y, float* norm_out,
                                       int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                                       const float* addend, int round_out, void* stream) {
  return ring_bf16_launch(x, wpack, bias, y, norm_out, N, D, H, W, he, flags, slope, eps, addend, round_out, 0, stream);
}

extern "C" int lf_conv3d_c16_ring_bf16_io(const void* x, const void* wpack, const float* bias, void* y, float* norm_out,
                                          int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                                          const void* addend, int round_out, int io, void* stream) {
  if (io < 0 || io > 7 || (addend == nullptr && (io & LF_IO_ADDEND_BF16))) return LF_EINVAL;
  return ring_bf16_launch(x, wpack, bias, y, norm_out, N, D, H, W, he, flags, slope, eps, addend, round_out, io, stream);
}

static int ring_bf16_launch(const void* x, const void* wpack, const float* bias, void* y, float* norm_out,
                            int N, int D, int H, int W, float he, unsigned flags, float slope, float eps,
                            const void* addend, int round_out, int io, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || round_out < 0 || round_out > 1) return LF_EINVAL;
  if ((long)D * H * W * 64 >= 0x7fffffffL || !(slope > 0.f && slope < 1.f)) return LF_EINVAL;
  if (flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(y) || !lf_aligned16(wpack) || (bias && !lf_aligned16(bias)) || (addend && !lf_aligned16(addend))) return LF_EALIGN;
  if (addend != nullptr && (flags != 0 || bias != nullptr || round_out != 0)) return LF_EINVAL;
  const int ptx = (W + TXs - 1) / TXs, pty = (H + TYs - 1) / TYs, ptz = (D + TZs - 1) / TZs;
  const long pt = (long)ptx * pty * ptz * N;
  if (pt > 0x7fffffffL) return LF_EINVAL;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const size_t shmem = (size_t)(LO_OFF + GUARD_B);
  typedef void (*kern_t)(const float*, const void*, const float*, float*, float*, int, int, int, int, int, int, int, int, float,
                         unsigned, float, float, const float*, const float*, unsigned, const float*, float*, int);
  // [addend form][io]: plain form = input / output storage (bits 0, 1); addend form = all three bits
  static const kern_t kerns[2][8] = {
      {conv3d_c16_f16x3_kernel<false, 1, 0>, conv3d_c16_f16x3_kernel<false, 1, 1>, conv3d_c16_f16x3_kernel<false, 1, 2>,
       conv3d_c16_f16x3_kernel<false, 1, 3>, nullptr, nullptr, nullptr, nullptr},
      {conv3d_c16_f16x3_kernel<true, 1, 0>, conv3d_c16_f16x3_kernel<true, 1, 1>, conv3d_c16_f16x3_kernel<true, 1, 2>,
       conv3d_c16_f16x3_kernel<true, 1, 3>, conv3d_c16_f16x3_kernel<true, 1, 4>, conv3d_c16_f16x3_kernel<true, 1, 5>,
       conv3d_c16_f16x3_kernel<true, 1, 6>, conv3d_c16_f16x3_kernel<true, 1, 7>}};
  const kern_t kern = kerns[addend != nullptr ? 1 : 0][io];
  if (kern == nullptr) return LF_EINVAL;
  const long want = (long)g_ring_bf16_wgs * cus;               // (35 KB of LDS and < 170 VGPRs: three fit)
  const unsigned grid = (unsigned)(pt < want ? pt : want);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shmem, (hipStream_t)stream, (const float*)x, wpack, bias, (float*)y, norm_out,
                     N, D, H, W, ptx, pty, ptz, (int)pt, he, flags, slope, eps, (const float*)addend, (const float*)nullptr,
                     (unsigned)LF_EPI_ADD, (const float*)nullptr, (float*)nullptr, round_out);
  return lf_launch_status();
}
