// Multi-output forms of the bf16 ring convolution (conv_split.hip, NP = 1) for the ConvGRU recurrence of the TRAINING step
// (BASELINE cfg 5; reference modules/gru.py:30-43 under recon/fusion.py:188-197, bf16 autocast policy):
//
//     u = s(conv_u([x, coords, h])),  r = s(conv_r([x, coords, h])),  c = conv_o([x, coords, h r]),  h' = h (1 - u) + c u
//
// A gate convolution over the concatenation is a sum of 16 -> 16 convolutions (ops_train._GruFuse), and several of them read
// the SAME 16-channel volume: x feeds three gates, h two, and in the backward pass every gate gradient is convolved with two
// transposed weight blocks (towards x and towards h).  The one-output ring kernel spends most of a tile's time outside its
// MFMAs (halo fetch, bf16 commit, barrier, epilogue: profiles/r05_ring_bf16_ablation.txt) -- here ONE staged halo serves NG
// output groups: a workgroup is NG x 4 waves, group g owns weight pack g, its own accumulators and its own epilogue, the
// halo pieces are fetched and committed by all NG x 4 waves together (6 / NG pieces each), the ring of six z-plane slots,
// the operand order and the MFMA order per accumulator are those of conv3d_c16_f16x3_kernel<*, 1> (same sums bit for bit).
// On top, the element-wise stages of the recurrence ride in the epilogue of the convolution that owns the same voxels (EX):
//     EX_RH    group 1 of (h -> upre, rpre): also r h = h s(rpre)                            (was lf_gru_train_stage_a)
//     EX_BLEND (r h -> cand): also h' = h (1 - s(upre)) + cand s(upre)                       (was lf_gru_train_stage_b)
//     EX_ABWD  group 0 of (gc -> g_rh, g_x): g_rh is not stored; grpre = g_rh h r (1 - r), gh12 = gh1 + g_rh r
//                                                                                          (was lf_gru_train_stage_a_bwd)
//     EX_BLOCK the Block step itself: bf16(bf16(conv) * he) + bias, LeakyReLU, PixelNorm (norms stored) -- the forward epilogue of
//              conv3d_c16_f16x3_kernel<false, 1> at compile time (1.63 -> 1.2 ms per 32 volumes)
//     EX_PREV  the data gradient of layer L+1 with the LeakyReLU' / PixelNorm' of layer L (the producer of its input) in the
//              store + L's bias-gradient sums: what comes out is L's PRE-activation gradient (was lf_epilogue_bwd_c16 on 32 views:
//              1.28 ms and a 4.3 GB round trip per Block of two 16 -> 16 layers)
// Epilogue forms per group (flags): result = fma(acc, he, addend) with an fp32 or bf16 addend (one volume for all samples
// or one per sample), or -- LF_RING_ROUND -- bf16(bf16(acc) * he), what autocast's half-precision convolution returns;
// stored as fp32 or bf16 records.  Sigmoids: 1 / (1 + 2^(-x log2 e)) on v_exp_f32 / v_rcp_f32 + one Newton step.
//
// Compiled with -fno-slp-vectorize (build.py), like conv_split.hip: with SLP vectorisation the epilogue's arithmetic becomes
// packed fp32 (v_pk_fma_f32 / v_pk_mul_f32), and the scheduler places a packed write of v[0:1] directly behind the
// buffer_store_dwordx4 v[0:3] of the previous output row; the compiler assumes no store-data hazard when the store's soffset
// is an SGPR (ours is: the z plane), the hardware still reads the data registers late -- the second half of the packed
// write reached the store for the last four lanes of every row of 16 (measured: element 1 of output rows 0-2, lanes 12-15,
// deterministic).  Scalar fp32 VALU code does not trigger it (tests/test_gru_ring_gpu.py pins every epilogue form).
#include "ring_tile.h"

namespace {

constexpr int DUMP_OFF = LO_OFF + GUARD_B;               // 512 B behind the ring that idle staging lanes write to
constexpr int LDSg = DUMP_OFF + 512;                     // 35,648 B

struct RmGroup { void* y; const void* add; unsigned flags; };
struct RmArgs {
  const void* x; const void* wpack;
  RmGroup g[2];
  const void* e0; const void* e1; void* o2; void* o3;
  int N, D, H, W, tiles_x, tiles_y, tiles_z, ntiles;
  float he;
  int add_per_sample;
  float slope;
};

__device__ __forceinline__ float sigmoid_fast(float v) {
  const float d = 1.f + __expf(-v);
  const float r = __builtin_amdgcn_rcpf(d);
  return r * (2.f - d * r);                                // (v_rcp_f32 + one Newton step)
}
__device__ __forceinline__ f32x4 sigmoid4_fast(const f32x4 v) {
  return (f32x4){sigmoid_fast(v[0]), sigmoid_fast(v[1]), sigmoid_fast(v[2]), sigmoid_fast(v[3])};
}
__device__ __forceinline__ f32x4 rb16x4(const f32x4 v) { return __builtin_convertvector(__builtin_convertvector(v, bf16x4s), f32x4); }

// FL >= 0 (one group only): the group's flags at compile time -- LF_RING_* | 8 = has an addend; the epilogue then is straight-line
// code between the MFMAs (runtime flags put branches around its loads and stores: measured 84 vs 59 us per volume).
template <int NG, bool IN16, int EX, int FL = -1>
__global__ void __launch_bounds__(256 * NG, 2) ring_multi_kernel(const RmArgs A) {
  static_assert(FL < 0 || NG == 1, "compile-time flags describe the only group");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int IN_SH = IN16 ? 5 : 6;                     // log2 bytes per input voxel record
  constexpr int NPW = NPIECE / NG;                        // halo pieces per wave and incoming plane pair
  constexpr int NT = 256 * NG;
  static_assert(NPIECE % NG == 0, "pieces divide over the groups");
  constexpr int OOB = (int)0x80000000;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = NG == 1 ? 0 : wv >> 2, w4 = wv & 3;     // output group; wave inside the group
  const int pz = w4 & 1, ry = w4 >> 1;                    // this wave's output plane and row quad
  const int n = lane & 15, kg = lane >> 4;
  const int D = A.D, H = A.H, W = A.W;
  for (int i = tid; i < LDSg / 16; i += NT) ((u32x4s*)smem)[i] = (u32x4s){0u, 0u, 0u, 0u};
  __syncthreads();

  const int nb = gridDim.x;
  const int lb = (nb % 8 == 0) ? (blockIdx.x % 8) * (nb / 8) + blockIdx.x / 8 : blockIdx.x;   // consecutive ranges per XCD
  const int per = (A.ntiles + nb - 1) / nb;
  const int t_begin = lb * per;
  const int t_end = min(t_begin + per, A.ntiles);
  if (t_begin >= t_end) return;

  const long nvox = (long)D * H * W;
  const unsigned sample_bytes = (unsigned)(nvox * 64);
  const float he = A.he;
  const RmGroup G = (NG > 1 && grp == 1) ? A.g[1] : A.g[0];
  const unsigned gfl = FL >= 0 ? (unsigned)FL : __builtin_amdgcn_readfirstlane(G.flags);
  const bool add16 = (gfl & LF_RING_ADD_BF16) != 0, out16 = (gfl & LF_RING_OUT_BF16) != 0, rnd = (gfl & LF_RING_ROUND) != 0;
  const bool has_add = FL >= 0 ? (FL & 8) != 0 : G.add != nullptr;
  const bool ex_rh = EX == 1 && grp == NG - 1;            // (wave-uniform)
  const bool ex_ab = EX == 3 && grp == 0;

  // ---- weights of this wave's group: A operand, lane (cout n, k = kg*8..+7) ----
  bf16x8s wreg[NPAIR];
  {
    const __bf16* wl = (const __bf16*)A.wpack + (long)grp * (NPAIR * 512) + n * 32 + kg * 8;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) wreg[p] = *(const bf16x8s*)(wl + p * 512);
  }

  // ---- halo pieces of this wave: of the six pieces of (incoming plane pz, rows 5 ry .. +4), pieces grp, grp + NG, ... ----
  int prow[NPW], pcol[NPW], plds[NPW];
  bool pok[NPW];
#pragma unroll
  for (int k = 0; k < NPW; ++k) {
    const int it = grp + k * NG;                          // (wave-uniform)
    if (it < 5) { prow[k] = 5 * ry + it; pcol[k] = lane >> 2; pok[k] = true; }
    else { prow[k] = 5 * ry + (lane >> 3); pcol[k] = 16 + ((lane >> 2) & 1); pok[k] = lane < 40; }
    plds[k] = (prow[k] * HXs + pcol[k]) * 32 + (lane & 3) * 8;
  }
  const int in_plane_bytes = (H * W) << IN_SH;
  const unsigned in_sample_bytes = (unsigned)(nvox << IN_SH);
  const int plane_bytes = H * W * 64;                     // fp32 records (halved per use for bf16 storage)

  int foff[NPW];
  const unsigned char* f_x = (const unsigned char*)A.x;
  auto fetch_column = [&](int bx, int by, int bn) {
    const int ox = bx * TXs - 1, oy = by * TYs - 1;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      const int col = ox + pcol[k], row = oy + prow[k];
      foff[k] = (pok[k] && (unsigned)col < (unsigned)W && (unsigned)row < (unsigned)H)
                    ? ((row * W + col) << IN_SH) + ((lane & 3) << (IN_SH - 2)) : OOB;
    }
    f_x = (const unsigned char*)A.x + ((long)bn * nvox << IN_SH);
  };
  u32x4s stg[NPW];
  auto fetch_plane = [&](int z, bool on) {
    const bool v = on && (unsigned)z < (unsigned)D;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)f_x, 0, v ? in_sample_bytes : 0u, 0x00020000);
    const int soff = v ? z * in_plane_bytes : 0;
#pragma unroll
    for (int k = 0; k < NPW; ++k) {
      if constexpr (IN16) {
        const u32x2s h = __builtin_amdgcn_raw_buffer_load_b64(rs, foff[k], soff, 0);
        stg[k] = (u32x4s){h[0], h[1], 0u, 0u};
      } else {
        stg[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, foff[k], soff, 0);
      }
    }
  };
  auto commit_piece = [&](int slot, auto kc) {
    constexpr int k = decltype(kc)::v;
    bf16x4s h;
    if constexpr (IN16) h = __builtin_bit_cast(bf16x4s, (u32x2s){stg[k][0], stg[k][1]});
    else h = __builtin_convertvector(__builtin_bit_cast(f32x4, stg[k]), bf16x4s);
    const int a = pok[k] ? slot * PLANE_B + plds[k] : DUMP_OFF + lane * 8;
    *(bf16x4s*)(smem + a) = h;
  };
  auto commit_plane = [&](int slot) { static_for<0, NPW>([&](auto kc) { commit_piece(slot, kc); }); };

  int cx, cy, cz, cn;
  {
    int tt = t_begin;                                     // z fastest: a workgroup walks up columns of tiles
    cz = tt % A.tiles_z; tt /= A.tiles_z;
    cx = tt % A.tiles_x; tt /= A.tiles_x;
    cy = tt % A.tiles_y; cn = tt / A.tiles_y;
  }
  int rot = 0;
  fetch_column(cx, cy, cn);
  fetch_plane(cz * TZs - 1 + pz, true);
  commit_plane(pz);
  fetch_plane(cz * TZs + 1 + pz, true);
  commit_plane(2 + pz);
  lds_barrier_s();

  const int lane_b = (4 * ry * HXs + n) * 32 + (kg & 1) * 16;

  // ---- epilogue of one tile (2 parts per output row, spread over the NEXT tile's MFMA phase) ----
  int eoff[RYs];
#pragma unroll
  for (int r = 0; r < RYs; ++r) eoff[r] = OOB;
  const unsigned char *e_y = nullptr, *e_add = nullptr, *e_x = nullptr, *e_e0 = nullptr, *e_e1 = nullptr, *e_o2 = nullptr, *e_o3 = nullptr;
  auto epi_column = [&](int bx, int by, int bn) {
    const int gx = bx * TXs + n, gy0 = by * TYs + RYs * ry;
#pragma unroll
    for (int r = 0; r < RYs; ++r) eoff[r] = (gx < W && gy0 + r < H) ? ((gy0 + r) * W + gx) * 64 + kg * 16 : OOB;
    const long sv = (long)bn * nvox;
    e_y = (const unsigned char*)G.y + (sv << (out16 ? 5 : 6));
    e_add = has_add ? (const unsigned char*)G.add + ((A.add_per_sample ? sv : 0L) << (add16 ? 5 : 6)) : nullptr;
    if constexpr (EX == 1) { e_x = (const unsigned char*)(A.e0 != nullptr ? A.e0 : A.x) + (sv << 6); e_o2 = (const unsigned char*)A.o2 + (sv << 5); }
    if constexpr (EX == 2) { e_e0 = (const unsigned char*)A.e0 + (sv << 6); e_e1 = (const unsigned char*)A.e1 + (sv << 5); e_o2 = (const unsigned char*)A.o2 + (sv << 6);
                             e_o3 = A.o3 != nullptr ? (const unsigned char*)A.o3 + (sv << 5) : nullptr; }
    if constexpr (EX == 3) { e_e0 = (const unsigned char*)A.e0 + (sv << 6); e_e1 = (const unsigned char*)A.e1 + (sv << 6); e_o2 = (const unsigned char*)A.o2 + (sv << 6); }
    if constexpr (EX == 5) { e_e0 = (const unsigned char*)A.e0 + (sv << 5); e_e1 = (const unsigned char*)A.e1 + (sv << 2); }
    if constexpr (EX == 4) e_o2 = (const unsigned char*)A.o2 + (sv << 2);
  };
  struct Epi { int soff; bool zv; };
  f32x4 ev[RYs], adv[RYs], x1[RYs], x2[RYs];
  f32x4 bacc = (f32x4){0.f, 0.f, 0.f, 0.f};                      // (EX_PREV) this lane's share of the producer's bias gradient
  f32x4 bias4 = (f32x4){0.f, 0.f, 0.f, 0.f};                     // (EX_BLOCK) bias of channels 4 kg .. +3
  if constexpr (EX == 4) { if (A.e0 != nullptr) bias4 = *(const f32x4*)((const float*)A.e0 + kg * 4); }
  float etv[RYs];
  auto rsrc = [&](const void* p, bool zv, bool half) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (zv && p != nullptr) ? (half ? sample_bytes >> 1 : sample_bytes) : 0u, 0x00020000);
  };
  auto ld32 = [&](const void* p, bool zv, int off, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc(p, zv, false), off, soff, 0));
  };
  auto ld16 = [&](const void* p, bool zv, int off, int soff) {   // (an out-of-volume offset 0x80000000 >> 1 is still past every buffer)
    return __builtin_convertvector(__builtin_bit_cast(bf16x4s, __builtin_amdgcn_raw_buffer_load_b64(rsrc(p, zv, true), off >> 1, soff >> 1, 0)), f32x4);
  };
  auto st32 = [&](const void* p, bool zv, int off, int soff, const f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, v), rsrc(p, zv, false), off, soff, 2);
  };
  auto st16 = [&](const void* p, bool zv, int off, int soff, const f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2s, __builtin_convertvector(v, bf16x4s)), rsrc(p, zv, true), off >> 1, soff >> 1, 2);
  };
  auto epi_tile = [&](Epi& E, int bz, bool valid) {      // the loads of the finished tile's epilogue, requested first
    const int gz = bz * TZs + pz;
    E.zv = valid && gz < D;
    E.soff = E.zv ? gz * plane_bytes : 0;
#pragma unroll
    for (int r = 0; r < RYs; ++r) {
      if (has_add) adv[r] = add16 ? ld16(e_add, E.zv, eoff[r], E.soff) : ld32(e_add, E.zv, eoff[r], E.soff);
      if constexpr (EX == 1) { if (ex_rh) x1[r] = ld32(e_x, E.zv, eoff[r], E.soff); }
      if constexpr (EX == 2) { x1[r] = ld32(e_e0, E.zv, eoff[r], E.soff); x2[r] = ld16(e_e1, E.zv, eoff[r], E.soff); }
      if constexpr (EX == 3) { if (ex_ab) { x1[r] = ld32(e_e0, E.zv, eoff[r], E.soff); x2[r] = ld32(e_e1, E.zv, eoff[r], E.soff); } }
      if constexpr (EX == 5) {
        x1[r] = ld16(e_e0, E.zv, eoff[r], E.soff);                 // the producer's activation (bf16 record quarter)
        const __amdgpu_buffer_rsrc_t rn = __builtin_amdgcn_make_buffer_rsrc((void*)e_e1, 0, E.zv ? (sample_bytes >> 4) : 0u, 0x00020000);
        x2[r][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rn, (eoff[r] >> 4) & ~3, E.soff >> 4, 0));   // its norm
      }
    }
  };
  auto epi_part = [&](const Epi& E, const f32x4 (&a)[RYs], auto rc, auto pc) {
    constexpr int r = decltype(rc)::v, part = decltype(pc)::v;
    if constexpr (part == 0) {
      f32x4 v;
      if (rnd) {
        v = rb16x4(rb16x4(a[r]) * he);                   // autocast: half conv result, `* he` in half
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = has_add ? __builtin_fmaf(a[r][e], he, adv[r][e]) : a[r][e] * he;
      }
      if constexpr (EX == 4) {
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float u = v[e] + bias4[e];
          u = fmaxf(u, u * A.slope);
          v[e] = u;
          ss += u * u;
        }
        etv[r] = quarter_sum(ss) * (1.f / 16.f) + 1e-8f;
      }
      ev[r] = v;
    } else {
      const f32x4 v = ev[r];
      if constexpr (EX == 4) {
        const float rinv = fast_rsqrt_s(etv[r]);
        const float rn = etv[r] * rinv;
        st16(e_y, E.zv, eoff[r], E.soff, v * rinv);
        const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc((void*)e_o2, 0, E.zv ? (sample_bytes >> 4) : 0u, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, rn), rsn, (eoff[r] >> 4) | (kg == 0 ? 0 : OOB), E.soff >> 4, 0);
        return;
      }
      if constexpr (EX == 3) {
        if (ex_ab) {                                     // v = g_rh (bf16-exact); adv = rpre, x1 = h, x2 = gh1
          const f32x4 rr = sigmoid4_fast(adv[r]);
          st16(e_y, E.zv, eoff[r], E.soff, v * x1[r] * rr * (1.f - rr));
          st32(e_o2, E.zv, eoff[r], E.soff, x2[r] + v * rr);
          return;
        }
      }
      if constexpr (EX == 5) {                                  // v = the gradient w.r.t. the producer's OUTPUT (bf16-exact)
        const f32x4 yp = x1[r];
        const float dot = quarter_sum(v[0] * yp[0] + v[1] * yp[1] + v[2] * yp[2] + v[3] * yp[3]) * (1.f / 16.f);
        const float rinv = fast_rcp_s(x2[r][0]);
        f32x4 gq;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float t = (v[e] - yp[e] * dot) * rinv;
          gq[e] = yp[e] > 0.f ? t : t * A.slope;
        }
        st16(e_y, E.zv, eoff[r], E.soff, gq);
        if (E.zv && eoff[r] != OOB) bacc += gq;                   // bias gradient of the producer: sums of the un-rounded values
        return;
      }
      if (out16) st16(e_y, E.zv, eoff[r], E.soff, v);
      else st32(e_y, E.zv, eoff[r], E.soff, v);
      if constexpr (EX == 1) {
        if (ex_rh) st16(e_o2, E.zv, eoff[r], E.soff, x1[r] * sigmoid4_fast(rb16x4(v)));   // r from rpre AS STORED (bf16)
      }
      if constexpr (EX == 2) {                           // v = cand; x1 = h, x2 = upre (as stored)
        const f32x4 uu = sigmoid4_fast(x2[r]), c = rb16x4(v);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = __fadd_rn(__fmul_rn(x1[r][e], __fsub_rn(1.f, uu[e])), __fmul_rn(c[e], uu[e]));
        st32(e_o2, E.zv, eoff[r], E.soff, o);
        if (e_o3 != nullptr) st16(e_o3, E.zv, eoff[r], E.soff, o);     // (a bf16 copy of the new state: what the next step's convolutions stage)
      }
    }
  };

  f32x4 accP[RYs];
#pragma unroll
  for (int r = 0; r < RYs; ++r) accP[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int px_ = cx, py_ = cy, pz_ = cz, pn_ = cn;
#ifndef RM_E0
#define RM_E0 1
#endif
  constexpr int E0 = RM_E0, PF = 2;                     // (RM_E0 >= 32: the finished tile's epilogue BEHIND the MFMA phase, A/B)
  for (int t = t_begin; t < t_end; ++t) {
    int nx = cx, ny = cy, nz = cz + 1, nn = cn;
    if (nz == A.tiles_z) { nz = 0; ++nx; }
    if (nx == A.tiles_x) { nx = 0; ++ny; }
    if (ny == A.tiles_y) { ny = 0; ++nn; }
    const bool on = t + 1 < t_end;
    const bool slide = on && nz != 0;
    if (t == t_begin + 1 || (t > t_begin && pz_ == 0)) epi_column(px_, py_, pn_);
    if (on && nz == 0) fetch_column(nx, ny, nn);
    Epi E;
    epi_tile(E, pz_, t > t_begin);
    fetch_plane(nz * TZs - 1 + (slide ? 2 : 0) + pz, on);
    __builtin_amdgcn_sched_barrier(0);

    const int s0 = mod6(rot + pz), s1 = mod6(s0 + 1), s2 = mod6(s1 + 1);
    const int aP = ((kg >> 1) ? s1 : s0) * PLANE_B + lane_b;
    const int aQ = s2 * PLANE_B + lane_b + (kg >> 1) * 32;
    const int aR = s2 * PLANE_B + lane_b + 2 * 32 + (kg >> 1) * (HXs * 32);
    const int cslot = mod6(rot + 4 + pz);

    f32x4 acc[RYs];
#pragma unroll
    for (int r = 0; r < RYs; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      bf16x8s bh[PF + 1];
      static_for<0, NOP + PF>([&](auto ic) {
        constexpr int i = decltype(ic)::v;
        if constexpr (i < NOP) {
          constexpr int cls = op_cls(i), h = op_h(i), kx = op_kx(i);
          const int addr = cls == 0 ? aP : (cls == 1 ? aQ : aR);
          constexpr int imm = (h * HXs + kx) * 32;
          bh[i % (PF + 1)] = *(const bf16x8s*)(smem + addr + imm);
        }
        if constexpr (i >= PF) {
          constexpr int j = i - PF;
          constexpr int cls = op_cls(j), h = op_h(j), kx = op_kx(j);
          const bf16x8s vh = bh[j % (PF + 1)];
          static_for<0, 3>([&](auto uc) {
            constexpr int u = decltype(uc)::v;
            constexpr int r = cls < 2 ? h - u : (u == 0 ? h : (u == 1 ? h - 2 : -1));
            constexpr int p = cls == 0 ? kx * 3 + u : (cls == 1 ? 9 + u : 12 + u);
            if constexpr (r >= 0 && r < RYs) acc[r] = mfma_k32(wreg[p], vh, acc[r]);
          });
        }
        if constexpr (i >= E0 && i < E0 + 2 * RYs) epi_part(E, accP, IC<(i - E0) / 2>{}, IC<(i - E0) % 2>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    if constexpr (E0 >= NOP + PF)
      static_for<0, 2 * RYs>([&](auto ic) { epi_part(E, accP, IC<decltype(ic)::v / 2>{}, IC<decltype(ic)::v % 2>{}); });
    commit_plane(cslot);
    lds_barrier_s();
    if (on && !slide) {
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
  {
    if (t_end - t_begin == 1 || pz_ == 0) epi_column(px_, py_, pn_);
    Epi E;
    epi_tile(E, pz_, true);
    static_for<0, 2 * RYs>([&](auto ic) { epi_part(E, accP, IC<decltype(ic)::v / 2>{}, IC<decltype(ic)::v % 2>{}); });
  }
  if constexpr (EX == 5) {
    // per-workgroup partial of the producer's bias gradient: lanes of one channel quad (kg) over the 16 voxels n, then the
    // four waves through LDS in wave order; partial[blockIdx][16] behind the 16 output floats of A.o2
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bacc[e] += __shfl_xor(bacc[e], o, 64);
    }
    __syncthreads();
    f32x4* red = (f32x4*)smem;
    if (n == 0) red[wv * 4 + kg] = bacc;
    __syncthreads();
    if (tid < 4) {
      const f32x4 t = (red[tid] + red[4 + tid]) + (red[8 + tid] + red[12 + tid]);
      *(f32x4*)((float*)A.o2 + 16 + (long)blockIdx.x * 16 + tid * 4) = t;
    }
  }
}

// out[0..15] = sum over the workgroups' partials (workgroup order, fp64)
__global__ void __launch_bounds__(64) bias_partials_sum_kernel(float* __restrict__ buf, int nblk) {
  const int c = threadIdx.x;
  if (c >= 16) return;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)buf[16 + (long)b * 16 + c];
  buf[c] = (float)s;
}

}  // namespace

static int ring_multi_launch(const void* x, int x_bf16, const void* wpack, int ngroups,
                             void* y0, const void* add0, unsigned flags0, void* y1, const void* add1, unsigned flags1,
                             int extra, const void* e0, const void* e1, void* o2, void* o3,
                             int N, int D, int H, int W, float he, int addend_per_sample, void* stream);

extern "C" int lf_conv3d_c16_ring_multi(const void* x, int x_bf16, const void* wpack, int ngroups,
                                        void* y0, const void* add0, unsigned flags0, void* y1, const void* add1, unsigned flags1,
                                        int extra, const void* e0, const void* e1, void* o2,
                                        int N, int D, int H, int W, float he, int addend_per_sample, void* stream) {
  return ring_multi_launch(x, x_bf16, wpack, ngroups, y0, add0, flags0, y1, add1, flags1, extra, e0, e1, o2, nullptr, N, D, H, W, he,
                           addend_per_sample, stream);
}

extern "C" int lf_conv3d_c16_ring_blend(const void* rh, const void* wpack, void* cand, const void* addend, const float* h, const void* upre,
                                        float* h_new, void* h_new_bf16, int N, int D, int H, int W, float he, void* stream) {
  return ring_multi_launch(rh, 1, wpack, 1, cand, addend, LF_RING_ADD_BF16 | LF_RING_OUT_BF16, nullptr, nullptr, 0, LF_RING_EX_BLEND, h, upre,
                           h_new, h_new_bf16, N, D, H, W, he, 1, stream);
}

static int ring_multi_launch(const void* x, int x_bf16, const void* wpack, int ngroups,
                             void* y0, const void* add0, unsigned flags0, void* y1, const void* add1, unsigned flags1,
                             int extra, const void* e0, const void* e1, void* o2, void* o3,
                             int N, int D, int H, int W, float he, int addend_per_sample, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || ngroups < 1 || ngroups > 2 || extra < 0 || extra > 5) return LF_EINVAL;
  if ((long)D * H * W * 64 >= 0x7fffffffL) return LF_EINVAL;
  if (x == nullptr || wpack == nullptr || y0 == nullptr || (ngroups == 2 && y1 == nullptr)) return LF_EINVAL;
  const unsigned known = LF_RING_ADD_BF16 | LF_RING_OUT_BF16 | LF_RING_ROUND;
  if ((flags0 & ~known) || (flags1 & ~known)) return LF_EINVAL;
  // the instantiated combinations (the launches of ops_train._GruFuse); everything else is refused
  typedef void (*kern_t)(const RmArgs);
  kern_t kern = nullptr;
  const int fl = (int)flags0 | (add0 != nullptr ? 8 : 0);
  if (ngroups == 2) {
    if (extra == LF_RING_EX_NONE && x_bf16) kern = ring_multi_kernel<2, true, 0>;
    else if (extra == LF_RING_EX_RH && !x_bf16) kern = ring_multi_kernel<2, false, 1>;
    else if (extra == LF_RING_EX_ABWD && x_bf16) kern = ring_multi_kernel<2, true, 3>;
  } else if (extra == LF_RING_EX_NONE && x_bf16) {
    switch (fl) {
      case 8: kern = ring_multi_kernel<1, true, 0, 8>; break;      // fp32 addend -> fp32 (the state's gradient)
      case 10: kern = ring_multi_kernel<1, true, 0, 10>; break;    // fp32 addend -> bf16 (the gates' view parts over the coordinate part)
      case 11: kern = ring_multi_kernel<1, true, 0, 11>; break;    // bf16 addend -> bf16 (the views' gradient, accumulated)
      case 6: kern = ring_multi_kernel<1, true, 0, 6>; break;      // rounded -> bf16
      default: kern = ring_multi_kernel<1, true, 0>;
    }
  } else if (extra == LF_RING_EX_NONE) {
    kern = fl == 11 ? ring_multi_kernel<1, false, 0, 11> : ring_multi_kernel<1, false, 0>;
  } else if (extra == LF_RING_EX_RH && !x_bf16) {
    kern = fl == 11 ? ring_multi_kernel<1, false, 1, 11> : ring_multi_kernel<1, false, 1>;
  } else if (extra == LF_RING_EX_RH && x_bf16 && fl == 11 && e0 != nullptr) {
    kern = ring_multi_kernel<1, true, 1, 11>;                      // (x = the bf16 copy of h; e0 = h itself for r h)
  } else if (extra == LF_RING_EX_BLEND && x_bf16) {
    kern = fl == 11 ? ring_multi_kernel<1, true, 2, 11> : ring_multi_kernel<1, true, 2>;
  } else if (extra == LF_RING_EX_ABWD && x_bf16) {
    kern = fl == 15 ? ring_multi_kernel<1, true, 3, 15> : ring_multi_kernel<1, true, 3>;
  } else if (extra == LF_RING_EX_PREV && x_bf16 && fl == 6) {
    kern = ring_multi_kernel<1, true, 5, 6>;
  } else if (extra == LF_RING_EX_BLOCK && fl == 6) {
    kern = x_bf16 ? ring_multi_kernel<1, true, 4, 6> : ring_multi_kernel<1, false, 4, 6>;
  }
  if (kern == nullptr) return LF_EINVAL;
  const unsigned flags_last = ngroups == 2 ? flags1 : flags0;
  if (extra == LF_RING_EX_RH && (o2 == nullptr || !(flags_last & LF_RING_OUT_BF16))) return LF_EINVAL;
  if (extra == LF_RING_EX_BLEND && (e0 == nullptr || e1 == nullptr || o2 == nullptr || !(flags0 & LF_RING_OUT_BF16))) return LF_EINVAL;
  if (extra == LF_RING_EX_BLOCK && (o2 == nullptr || ngroups != 1)) return LF_EINVAL;
  if (extra == LF_RING_EX_PREV && (e0 == nullptr || e1 == nullptr || o2 == nullptr || ngroups != 1)) return LF_EINVAL;
  if (extra == LF_RING_EX_ABWD && (e0 == nullptr || e1 == nullptr || o2 == nullptr || add0 == nullptr ||
                                   !(flags0 & LF_RING_ROUND) || !(flags0 & LF_RING_ADD_BF16))) return LF_EINVAL;
  const void* ps[] = {x, wpack, y0, add0, y1, add1, e0, e1, o2, o3};
  for (const void* p : ps)
    if (p != nullptr && !lf_aligned16(p)) return LF_EALIGN;
  RmArgs A;
  A.x = x; A.wpack = wpack;
  A.g[0] = RmGroup{y0, add0, flags0};
  A.g[1] = RmGroup{y1, add1, flags1};
  A.e0 = e0; A.e1 = e1; A.o2 = o2; A.o3 = o3;
  A.N = N; A.D = D; A.H = H; A.W = W;
  A.tiles_x = (W + TXs - 1) / TXs; A.tiles_y = (H + TYs - 1) / TYs; A.tiles_z = (D + TZs - 1) / TZs;
  const long pt = (long)A.tiles_x * A.tiles_y * A.tiles_z * N;
  if (pt > 0x7fffffffL) return LF_EINVAL;
  A.ntiles = (int)pt;
  A.he = he;
  A.add_per_sample = addend_per_sample ? 1 : 0;
  A.slope = 0.2f;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  const long want = (long)(ngroups == 1 ? 2 : 1) * cus;           // 8 waves per CU either way
  const unsigned grid = (unsigned)(pt < want ? pt : want);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * ngroups), (size_t)LDSg, (hipStream_t)stream, A);
  if (extra == LF_RING_EX_PREV) {
    const int st = lf_launch_status();
    if (st) return st;
    hipLaunchKernelGGL(bias_partials_sum_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (float*)o2, (int)grid);
  }
  return lf_launch_status();
}
