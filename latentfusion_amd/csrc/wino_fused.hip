// Wide Winograd convolutions, stage 2 + 3 in one kernel (gfx950): the per-frequency products
//   M[f] = V[f] (tiles x Cin) . U[f] (Cin x Cout),   f = 0 .. F-1   (F = 64 for F(2x2x2,3x3x3), 16 for F(2x2,3x3))
// on the fp32 MFMA (v_mfma_f32_16x16x4_f32, exact fp32 products and accumulation) with the OUTPUT TRANSFORM
// Y = A^T M A (x, y[, z]) folded into the frequency loop and the conv epilogue (He scale, bias, LeakyReLU) into the
// store -- the Winograd-domain products M (8x / 4x the size of the output) never exist in memory, and no library GEMM
// sits on the hot path of the released-width model (latentfusion/modules/blocks.py:152-158 at 64-512 channels,
// tools/train/train.sh:28-66; the same kernel evaluates the data gradients with transposed / flipped weight packs).
//
// Work split: a 256-thread workgroup owns 64 output channels x 64 Winograd tiles for ALL frequencies; wave (wr, wc)
// owns the 32 x 32 block (couts wr*32.., tiles wc*32..) as 2 x 2 MFMA tiles.  MFMA roles are swapped w.r.t. a textbook
// GEMM -- A = weights [16 couts][k], B = inputs [k][16 tiles] -- so a lane ends up with 4 CONSECUTIVE output channels
// of one tile and the channels-last store is a float4 per lane (64 B contiguous per 4 lanes).
// The (f, k) loop is flattened into stages of KC = 32 input channels: the A (64 x 32) and B (64 x 32) chunks of a
// stage are DMA'd global -> LDS (buffer_load_dwordx4 ... lds) into a ring of 4 stages, three stages ahead of the one
// that feeds the 32 MFMAs per wave, with ONE workgroup barrier per stage (measured: with a register-staged double
// buffer, one stage ahead, the kernel ran at ~15 % of the fp32 MFMA peak -- bound by the latency of each stage's
// loads); LDS rows are 128 B with the eight 16-byte chunks XOR-swizzled by (row >> 1) & 7, which makes the hardware's
// 16-lane ds_read_b128 groups hit distinct banks.  Within a 16-wide k-group a
// lane group kg = lane >> 4 takes k = kg*4 + i in MFMA step i (the contraction index may be permuted freely as long as
// A and B agree), so one ds_read_b128 feeds four MFMA steps.
// After the last chunk of a frequency the 2 x 2 accumulators are folded into the 2^dims output accumulators with the
// (wave-uniform) coefficients A^T[o][f] in {-1, 0, 1}.
#include "lf_common.h"

namespace {

constexpr int KC = 32;      // input channels per stage
constexpr int NSTAGE = 4;   // LDS ring depth (stages of A + B chunks); a power of two

typedef unsigned u32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// byte offset of 16-byte chunk c (0..7) of row r in a rows x 32-float LDS tile
__device__ __forceinline__ int lds_chunk(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

// A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]: coefficient of frequency component a in output o
__device__ __forceinline__ float at_coef(int o, int a) {
  return o == 0 ? (a < 3 ? 1.f : 0.f) : (a == 0 ? 0.f : (a == 1 ? 1.f : -1.f));
}

// Workgroup shape: WM x WN waves, each owning BA x BB MFMA blocks (16 couts x 16 tiles each):
//   NT = WM*BA*16 output channels (A rows) x MT = WN*BB*16 Winograd tiles (B rows) per workgroup.
// The global -> LDS bytes per MFMA fall with the block area -- 64 x 64: (64+64)*128 B per stage for 32 MFMAs per wave and
// four waves = 16 B per CU clock at the full fp32 MFMA rate, which is all the vector-memory path of a CU delivers
// (measured: 0.50 of the MFMA peak); 128 x 64: 12 B/clk; 128 x 128: 8 B/clk -- but the 2^dims output accumulators per
// block bound the blocks per wave (3-D: 36 VGPRs per block).
template <int DIMS, int WM, int WN, int BA, int BB>
__global__ void __launch_bounds__(WM * WN * 64, (WM * WN == 4 ? 2 : 1)) wino_fused_kernel(
    const float* __restrict__ V, const float* __restrict__ U2, const float* __restrict__ bias, float* __restrict__ y,
    long T, int tz, int ty, int tx, int D, int H, int W, int Cin, int Cout, int CoutP, float he, unsigned flags, float slope,
    float* __restrict__ partial, long ysize) {
  constexpr int F = DIMS == 3 ? 64 : 16;
  constexpr int NO = DIMS == 3 ? 8 : 4;                          // outputs per tile
  constexpr int NTc = WM * BA * 16, MTc = WN * BB * 16, NW = WM * WN;
  constexpr int A_BYTES = NTc * 128, STAGE_BYTES = (NTc + MTc) * 128;
  constexpr int PA = NTc / 8, PB = MTc / 8;                      // 1 KiB DMA pieces of the A / B chunk of a stage
  constexpr int PPW = (PA + PB) / NW;                            // pieces per wave and stage
  static_assert((PA + PB) % NW == 0 && PPW <= 8, "pieces must split evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w / WN, wc = w % WN;
  const int lr = lane & 15, kg = lane >> 4;
  // (round 6) XCD-aware order: the workgroups of ONE tile block -- one per output-channel block, all reading the same slice of V --
  // are dispatched x-fastest, i.e. gridDim.x dispatches apart, and each re-read V from HBM (9.8 GB per 128-render launch of the
  // released architecture against 4.8 GB of V + y).  Dispatch L goes to XCD L % 8: re-numbered so that the channel blocks of a tile
  // block follow each other ON ONE XCD, they stream V through that XCD's L2 together.
  int bxi = blockIdx.x, byi = blockIdx.y;
#ifndef WF_XCD
#define WF_XCD 1
#endif
  if (WF_XCD && gridDim.y > 1 && (gridDim.x & 7) == 0) {
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y, slot = L >> 3;
    byi = (int)(slot % gridDim.y);
    bxi = (int)((slot / gridDim.y) * 8 + (L & 7));
  }
  const long m0 = (long)bxi * MTc;                               // first tile of this workgroup
  const int n0 = byi * NTc;                                      // first output channel

  // ---- global -> LDS staging by LDS-DMA (buffer_load_dwordx4 ... lds): a wave-instruction deposits 64 x 16 B = 1 KiB
  // linearly at a wave-uniform LDS address, so the swizzle is applied on the GLOBAL side: the lane that lands on
  // LDS position pos = piece*64 + lane (row r = pos >> 3, slot pos & 7) fetches logical chunk c = slot ^ ((r >> 1) & 7)
  // of that row.  A stage = PA pieces of A + PB of B; wave w issues pieces w*PPW .. +PPW-1 of the concatenated list.
  // No staging registers, no ds_write; out-of-range chunks (k >= Cin, tile >= T, cout >= CoutP) get an out-of-range
  // offset and the DMA writes zeros. ----
  const u32 slabV = (u32)((long)T * Cin * 4 <= 0xffffffffL ? (long)T * Cin * 4 : 0xffffffffL);
  const u32 slabU = (u32)((long)CoutP * Cin * 4);
  int voff[PPW], kch[PPW], ldso[PPW];
  bool isA[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int p = w * PPW + i;                                   // wave-uniform
    isA[i] = p < PA;
    const int piece = isA[i] ? p : p - PA;
    const int pos = piece * 64 + lane, r = pos >> 3, c = (pos & 7) ^ ((r >> 1) & 7);
    kch[i] = c * 4;
    if (isA[i]) {
      const long off = (long)(n0 + r) * Cin * 4 + c * 16;        // U2[f][n0 + r][k0 + 4c ..]
      voff[i] = off < (long)slabU ? (int)(u32)off : 0x7fffffff;
      ldso[i] = piece * 1024;
    } else {
      const long row = m0 + r;
      voff[i] = row < T ? (int)((u32)row * (u32)Cin * 4u + (u32)c * 16u) : 0x7fffffff;
      ldso[i] = A_BYTES + piece * 1024;
    }
  }
  const int nk = (Cin + KC - 1) / KC;
  // frequency split (small problems: few tile / channel blocks): workgroup z handles frequencies
  // [z * F / gridDim.z, (z + 1) * F / gridDim.z) and writes its un-scaled partial outputs; lf's finish kernel adds the
  // partials in a fixed order and applies the epilogue
  const int fper = F / gridDim.z, f_first = blockIdx.z * fper;
  const int S = fper * nk;
  // The issue side keeps its own cursor (frequency, k-chunk, ring slot) three stages ahead of the compute side, so a
  // piece costs one LDS-DMA instruction and no address arithmetic: the per-lane byte offset inside the frequency slab
  // is a loop invariant (voffset), the k-chunk advances through the instruction's SCALAR offset, and the two buffer
  // descriptors are rebuilt only when the cursor enters the next frequency.  (First version: stage index -> (f, k) by
  // division and fresh descriptors per piece = 4.6 scalar instructions per MFMA, MFMA pipe 54 % busy.)
  const long strideU = (long)CoutP * Cin, strideV = (long)T * Cin;
  const float* pU = U2 + (long)f_first * strideU;
  const float* pV = V + (long)f_first * strideV;
  __amdgpu_buffer_rsrc_t ru = __builtin_amdgcn_make_buffer_rsrc((void*)pU, 0, slabU, 0x00020000);
  __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)pV, 0, slabV, 0x00020000);
  int ik = 0, islot = 0, issued = 0;                             // cursor: k-chunk of the stage being issued, its ring slot
  const bool ktail = (Cin % KC) != 0;
  // piece q (0 .. PPW-1) of the cursor's stage for this wave
  auto issue_piece = [&](int q) {
    unsigned char* slot = smem + islot * STAGE_BYTES;
    const int k0b = ik * KC * 4;                                 // scalar byte offset of the k-chunk
    int vo = voff[q];
    if (ktail && ik == nk - 1) vo = (ik * KC + kch[q] < Cin) ? vo : 0x7fffffff;   // ragged last chunk: lanes beyond Cin read zeros
    if (isA[q])
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ru, (__attribute__((address_space(3))) void*)(slot + ldso[q]), 16, vo, k0b, 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rv, (__attribute__((address_space(3))) void*)(slot + ldso[q]), 16, vo, k0b, 0, 0);
    if (q == PPW - 1) {                                          // stage complete: advance the cursor
      ++issued;
      islot = (islot + 1) & (NSTAGE - 1);
      if (++ik == nk) {
        ik = 0;
        pU += strideU;
        pV += strideV;
        ru = __builtin_amdgcn_make_buffer_rsrc((void*)pU, 0, slabU, 0x00020000);
        rv = __builtin_amdgcn_make_buffer_rsrc((void*)pV, 0, slabV, 0x00020000);
      }
    }
  };

  // ---- MFMA operand addressing: row of this lane in A (cout) and B (tile) for the 16-row blocks of the wave ----
  int rdA[BA][2], rdB[BB][2];                                    // [row block][k-group j]
#pragma unroll
  for (int j = 0; j < 2; ++j) {
#pragma unroll
    for (int t = 0; t < BA; ++t) rdA[t][j] = lds_chunk(wr * (BA * 16) + t * 16 + lr, j * 4 + kg);
#pragma unroll
    for (int t = 0; t < BB; ++t) rdB[t][j] = A_BYTES + lds_chunk(wc * (BB * 16) + t * 16 + lr, j * 4 + kg);
  }

  f32x4 Y[NO][BA][BB];
#pragma unroll
  for (int o = 0; o < NO; ++o)
#pragma unroll
    for (int a = 0; a < BA; ++a)
#pragma unroll
      for (int b = 0; b < BB; ++b) Y[o][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 acc[BA][BB];
#pragma unroll
  for (int a = 0; a < BA; ++a)
#pragma unroll
    for (int b = 0; b < BB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ring of NSTAGE stages: stages s+1 .. s+NSTAGE-1 are in flight while stage s feeds the MFMAs
  for (int s0 = 0; s0 < NSTAGE - 1 && s0 < S; ++s0)
#pragma unroll
    for (int q = 0; q < PPW; ++q) issue_piece(q);
  int kc = 0, f = f_first;
  for (int s = 0; s < S; ++s) {
    // this wave's pieces of stage s have landed when at most the pieces of the later stages it issued are outstanding
    // (PPW DMA instructions per stage and wave); then the workgroup barrier makes every wave's pieces visible AND
    // certifies that everybody is done reading stage s-1, whose ring slot the next issue overwrites
    const int ahead = min(NSTAGE - 2, S - 1 - s);
    if (ahead >= 2) __builtin_amdgcn_s_waitcnt(0x0f70 | ((2 * PPW) & 15) | (((2 * PPW) >> 4) << 14));     // vmcnt(2 PPW), the rest open
    else if (ahead == 1) __builtin_amdgcn_s_waitcnt(0x0f70 | (PPW & 15));                                  // vmcnt(PPW)
    else __builtin_amdgcn_s_waitcnt(0x0f70);                                                                // vmcnt(0)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const bool more = issued < S;                                // wave-uniform
    const unsigned char* base = smem + (s % NSTAGE) * STAGE_BYTES;
    f32x4 fa[2][BA], fb[2][BB];                                  // [k-group j][row block]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int t = 0; t < BA; ++t) fa[j][t] = *(const f32x4*)(base + rdA[t][j]);
#pragma unroll
      for (int t = 0; t < BB; ++t) fb[j][t] = *(const f32x4*)(base + rdB[t][j]);
    }
#if defined(WF_ABL_XFORM) && WF_ABL_XFORM
    // ABLATION (tools/wino_fused_ab.py; never in the shipped library): the instruction stream of forming every B fragment
    // from the 8 raw halo records of its Winograd frequency INSIDE this kernel -- 7 more ds_read_b128
    // and 7 packed add/sub pairs per fragment -- on top of the unchanged DMA of a same-sized chunk: an upper bound on what
    // an in-kernel input transform could reach (its halo staging would cost extra on top).  Results are garbage.
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int t = 0; t < BB; ++t)
#pragma unroll
        for (int r = 1; r < 8; ++r) {
          const f32x4 e = *(const f32x4*)(base + A_BYTES + ((rdB[t][j] - A_BYTES + r * 1040) & (MTc * 128 - 16)));
          fb[j][t] = (r & 1) ? fb[j][t] + e : fb[j][t] - e;
        }
#endif
    // the DMA pieces of stage s+3 are issued BETWEEN groups of MFMAs: an LDS-DMA instruction holds the issuing
    // (in-order) wave for ~60-180 cycles, which the matrix work already queued ahead of it covers
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int a = 0; a < BA; ++a)
#pragma unroll
          for (int b = 0; b < BB; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[j][a][i], fb[j][b][i], acc[a][b], 0, 0, 0);
        // issue slots: after every second k-step (<= 4 pieces per wave) or after every k-step (more)
        constexpr int SLOT_EVERY = PPW <= 4 ? 2 : 1;
        if ((i + 1) % SLOT_EVERY == 0) {
          const int slot_ = (j * 4 + i) / SLOT_EVERY;
          __builtin_amdgcn_sched_barrier(0);
          if (slot_ < PPW && more) issue_piece(slot_);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    if (++kc == nk) {
      // frequency f complete: fold it into the outputs, Y[o] += A^T[o][f] * M[f]
      kc = 0;
      const int fc = f & 3, fb_ = (f >> 2) & 3, fa_ = (f >> 4) & 3;      // x, y, z frequency (DIMS == 2: fa_ unused)
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        float cf = at_coef(o & 1, fc) * at_coef((o >> 1) & 1, fb_);
        if (DIMS == 3) cf *= at_coef((o >> 2) & 1, fa_);
        if (cf != 0.f) {                                         // wave-uniform
#pragma unroll
          for (int a = 0; a < BA; ++a)
#pragma unroll
            for (int b = 0; b < BB; ++b) Y[o][a][b] += acc[a][b] * cf;
        }
      }
#pragma unroll
      for (int a = 0; a < BA; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ++f;
    }
  }

  // ---- epilogue: He scale, bias, LeakyReLU; lane holds couts n0 + wr*BA*16 + a*16 + kg*4 .. +3 of tile column lr ----
#pragma unroll
  for (int b = 0; b < BB; ++b) {
    const long tile = m0 + wc * (BB * 16) + b * 16 + lr;
    if (tile >= T) continue;
    long r = tile;
    const int bx = (int)(r % tx); r /= tx;
    const int by = (int)(r % ty); r /= ty;
    const int bz = DIMS == 3 ? (int)(r % tz) : 0;
    const long n = DIMS == 3 ? r / tz : r;
#pragma unroll
    for (int a = 0; a < BA; ++a) {
      const int co = n0 + wr * (BA * 16) + a * 16 + kg * 4;
      if (co >= Cout) continue;
      f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (bias != nullptr) bv = *(const f32x4*)(bias + co);
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const int gx = 2 * bx + (o & 1), gy = 2 * by + ((o >> 1) & 1), gz = 2 * bz + (DIMS == 3 ? ((o >> 2) & 1) : 0);
        if (gx >= W || gy >= H || gz >= D) continue;
        // LF_OUT_DEPTH_INNER: y as [N][H][W][D][Cout] -- the factor projection then reads a pixel's D x Cout column as ONE row
        const long vox = (flags & LF_OUT_DEPTH_INNER) ? ((n * H + gy) * W + gx) * D + gz : ((n * D + gz) * H + gy) * W + gx;
        if (partial != nullptr) {                                // frequency-split launch: raw partial sums
          *(f32x4*)(partial + (long)blockIdx.z * ysize + vox * Cout + co) = Y[o][a][b];
          continue;
        }
        f32x4 v = Y[o][a][b] * he + bv;
        if (flags & LF_EPI_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
        }
        *(f32x4*)(y + vox * Cout + co) = v;
      }
    }
  }
}

// y = epilogue(he * sum_z partial[z] + bias): fixed-order sum of the frequency-split partial outputs
__global__ void __launch_bounds__(256) wino_fused_finish_kernel(const f32x4* __restrict__ partial, const float* __restrict__ bias,
                                                               f32x4* __restrict__ y, long n4, long ysize4, int zs, int c4,
                                                               float he, unsigned flags, float slope) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  f32x4 acc = partial[i];
  for (int z = 1; z < zs; ++z) acc += partial[i + z * ysize4];
  f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) bv = *(const f32x4*)(bias + (i % c4) * 4);
  f32x4 v = acc * he + bv;
  if (flags & LF_EPI_LRELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
  }
  y[i] = v;
}

// Workgroup shapes (NT output channels x MT tiles): 0 = 64 x 64 (4 waves, two workgroups per CU), 1 = 128 x 64,
// 2 = 64 x 128, 3 = 128 x 128, 4 = 64 x 256 (8 waves, one workgroup per CU; 3 and 4 are 2-D only: the 3-D kernel keeps 8 output
// accumulators per block)
struct FusedCfg { int nt, mt, waves; };
constexpr int NCFG = 5;
constexpr FusedCfg kCfg[NCFG] = {{64, 64, 4}, {128, 64, 8}, {64, 128, 8}, {128, 128, 8}, {64, 256, 8}};
int g_fused_cfg = -1;                                             // lf_set_tuning(3, v): -1 = pick by shape

int pick_fused_cfg(int dims, long T, int CoutP) {
  if (g_fused_cfg >= 0 && g_fused_cfg < NCFG && !(dims == 3 && g_fused_cfg >= 3)) return g_fused_cfg;
  // measured per layer shape of the released model at N = 8 / 32 / 128 (tools/wide_conv_probe.py --cfgs,
  // profiles/r02_wide_conv_cfgs_*.jsonl): the large shapes pay (+10..20 % on the GEMM stage) once there are >= 1-2 k tiles
  // and >= 128 output channels; 64-channel layers and tiny maps stay on 64 x 64 with two workgroups per CU
  if (CoutP >= 128 && dims == 2 && T >= 1024) return 3;
  if (CoutP < 128 && dims == 2 && T >= 200000) return 4;        // 64-channel layers on the largest maps: 64 x 256
  if (CoutP >= 128 && dims == 3 && T >= 32768) return 1;
  if (CoutP >= 128 && dims == 3 && T >= 2048) return 2;
  return 0;
}

// frequency split of a launch with gx x gy tile / channel blocks: enough workgroups to fill the chip
int fused_zsplit(int dims, long gx, int gy, int waves) {
  const int F = dims == 3 ? 64 : 16;
  const long want = waves == 4 ? 512 : 256;                      // two (4-wave) / one (8-wave) resident workgroups per CU
  int zs = 1;
  while (zs < F && gx * gy * zs < want) zs <<= 1;
  return zs;
}

}  // namespace

extern "C" int lf_wino_fused_cout_padded(int Cout) { return (Cout + 63) / 64 * 64; }

// bytes of scratch lf_wino_fused_gemm needs for this shape (0: none).  (An upper bound over the workgroup shapes.)
extern "C" size_t lf_wino_fused_scratch_bytes(int dims, int N, int D, int H, int W, int Cout) {
  if ((dims != 2 && dims != 3) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  const int tz = dims == 3 ? (D + 1) / 2 : 1, ty = (H + 1) / 2, tx = (W + 1) / 2;
  const long T = (long)N * tz * ty * tx;
  const int CoutP = lf_wino_fused_cout_padded(Cout);
  int zs = 1;
  for (int c = 0; c < NCFG; ++c) {
    if (dims == 3 && c >= 3) continue;
    const int z = fused_zsplit(dims, (T + kCfg[c].mt - 1) / kCfg[c].mt, (CoutP + kCfg[c].nt - 1) / kCfg[c].nt, kCfg[c].waves);
    zs = z > zs ? z : zs;
  }
  return zs > 1 ? (size_t)zs * N * D * H * W * Cout * sizeof(float) : 0;
}

template <int DIMS, int WM, int WN, int BA, int BB>
static int launch_fused(dim3 grid, hipStream_t s, const float* V, const float* U2, const float* bias, float* y, long T, int tz, int ty,
                        int tx, int D, int H, int W, int Cin, int Cout, int CoutP, float he, unsigned flags, float slope,
                        float* partial, long ysize) {
  constexpr int lds = NSTAGE * (WM * BA * 16 + WN * BB * 16) * 128;
  static lf_devmask_t attr_set;
  {
    hipError_t e = lf_ensure_dyn_lds(attr_set, (const void*)wino_fused_kernel<DIMS, WM, WN, BA, BB>, lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((wino_fused_kernel<DIMS, WM, WN, BA, BB>), grid, dim3(WM * WN * 64), lds, s, V, U2, bias, y, T, tz, ty, tx, D, H, W,
                     Cin, Cout, CoutP, he, flags, slope, partial, ysize);
  return lf_launch_status();
}

// y = epilogue(output_transform(V[f] . U2[f]^T)):  V [F][T][Cin] from lf_wino{2,3}d_input_transform;
// scratch: lf_wino_fused_scratch_bytes(...) bytes (small problems are split over the frequencies);
// U2 [F][CoutP][Cin] (output-channel major, CoutP = lf_wino_fused_cout_padded(Cout), zero padded);
// y channels-last [N][D][H][W][Cout].  flags: LF_EPI_LRELU (PixelNorm: run lf_pixelnorm_fwd on y afterwards).
extern "C" int lf_wino_fused_gemm(const float* V, const float* U2, const float* bias, float* y, void* scratch,
                                  size_t scratch_bytes, int dims, int N, int D, int H, int W, int Cin, int Cout, float he,
                                  unsigned flags, float slope, void* stream) {
  lf_clear_error();
  if ((dims != 2 && dims != 3) || N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return LF_EINVAL;
  if ((Cin & 3) || (Cout & 3) || (dims == 2 && D != 1) || (flags & ~(LF_EPI_LRELU | LF_OUT_DEPTH_INNER))) return LF_EINVAL;
  if (!lf_aligned16(V) || !lf_aligned16(U2) || !lf_aligned16(y) || (bias && !lf_aligned16(bias))) return LF_EALIGN;
  const int tz = dims == 3 ? (D + 1) / 2 : 1, ty = (H + 1) / 2, tx = (W + 1) / 2;
  const long T = (long)N * tz * ty * tx;
  const int CoutP = lf_wino_fused_cout_padded(Cout);
  // 32-bit byte offsets inside one frequency slab
  if (T * Cin * 4 > 0xffffffffL || (long)CoutP * Cin * 4 > 0xffffffffL) return LF_EINVAL;
  const int cfg = pick_fused_cfg(dims, T, CoutP);
  const int MT = kCfg[cfg].mt, NT = kCfg[cfg].nt;
  const long gx = (T + MT - 1) / MT;
  const int gy = (CoutP + NT - 1) / NT;
  if (gx > 0x7fffffffL || gy > 65535) return LF_EINVAL;
  const int zs = fused_zsplit(dims, gx, gy, kCfg[cfg].waves);
  const long ysize = (long)N * D * H * W * Cout;
  if (zs > 1 && (scratch == nullptr || scratch_bytes < (size_t)zs * ysize * sizeof(float) || !lf_aligned16(scratch))) return LF_ENOSPC;
  float* partial = zs > 1 ? (float*)scratch : nullptr;
  dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)zs);
  hipStream_t s = (hipStream_t)stream;
#define LF_FUSED(DIMS_, WM_, WN_, BA_, BB_) \
  launch_fused<DIMS_, WM_, WN_, BA_, BB_>(grid, s, V, U2, bias, y, T, tz, ty, tx, D, H, W, Cin, Cout, CoutP, he, flags, slope, partial, ysize)
  int st;
  if (dims == 3) st = cfg == 1 ? LF_FUSED(3, 4, 2, 2, 2) : (cfg == 2 ? LF_FUSED(3, 2, 4, 2, 2) : LF_FUSED(3, 2, 2, 2, 2));
  else st = cfg == 1 ? LF_FUSED(2, 4, 2, 2, 2) : (cfg == 2 ? LF_FUSED(2, 2, 4, 2, 2) : (cfg == 3 ? LF_FUSED(2, 4, 2, 2, 4) :
            (cfg == 4 ? LF_FUSED(2, 1, 8, 4, 2) : LF_FUSED(2, 2, 2, 2, 2))));
#undef LF_FUSED
  if (st || zs == 1) return st;
  const long n4 = ysize / 4;
  hipLaunchKernelGGL(wino_fused_finish_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const f32x4*)partial, bias, (f32x4*)y,
                     n4, n4, zs, Cout / 4, he, flags & LF_EPI_LRELU, slope);
  return lf_launch_status();
}

// tuning hook for lf_set_tuning (key 3, resample.hip): workgroup shape of the fused GEMM, -1 = by shape
int lf_internal_fused_set_cfg(int v) {
  const int prev = g_fused_cfg;
  if (v >= -1 && v < NCFG) g_fused_cfg = v;
  return prev;
}
