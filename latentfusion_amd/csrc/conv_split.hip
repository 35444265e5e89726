// Split-precision variant of the dominant kernel: conv3d 16->16 on the f16 matrix cores with
// fp32-equivalent accuracy ("f16x3").
//
// Every fp32 operand x is split as x = hi + lo with hi = (f16)x, lo = (f16)(x - hi): 22 mantissa bits.
// The product is formed from three f16 MFMAs accumulating in fp32,
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (dropped a_lo*b_lo <= 2^-22 |a b|),
// each partial product being exact in fp32 (11 x 11 significant bits).  The f16 MFMA runs at 16x the
// fp32 MFMA rate, so three of them cost 3/16 of the exact-fp32 kernel's matrix time and the kernel
// becomes HBM/LDS-bound instead of MFMA-bound.  Activations stay fp32 in HBM: the split happens
// on-chip, between the LDS-DMA'd fp32 halo tile and the f16 hi/lo tiles the MFMAs read.
//
// Range: forward activations are O(1) (PixelNorm output) and need no scaling.  Gradients can be tiny,
// so data-gradient launches multiply the input by a power of two derived from the tensor's max-abs
// (`amax_in`, produced by the previous kernel with an order-independent atomic max) and undo it
// exactly in the epilogue; elements far below the tensor max lose relative but not absolute accuracy,
// which is what a dot product needs.
//
// Structure (same skeleton as conv3d_c16_persistent_kernel in conv.hip): one 512-thread workgroup per
// CU walks 4x8x16 tiles; weights (hi and lo, 14 tap pairs x 32-deep K) live in registers; halo of
// tile t+1 is DMA'd (fp32) while tile t is multiplied; SIMD-partner waves skew their epilogues.
#include "lf_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TXs = 16, TYs = 8, TZs = 4, HXs = 18, HYs = 10, HZs = 6;
constexpr int HALOs = HZs * HYs * HXs;                   // 1080 voxels
constexpr int NSLOTs = (HALOs * 4 + 63) / 64;            // 68 DMA pieces (1 KiB each) per tile
constexpr int NITs = (NSLOTs + 7) / 8;
constexpr int RAW_FLOATS = NSLOTs * 256;                 // fp32 DMA target: 69,632 B
constexpr int HALF_ELEMS = HALOs * 16;                   // one f16 plane: 34,560 B
constexpr int NPAIR = 14;                                // 27 taps -> 13 pairs + 1 padded

// Tap pairing.  One K=32 MFMA consumes two taps (lanes kg 0,1: first tap; kg 2,3: second tap).  Pairs
// are chosen so that the second tap sits at one of only three constant voxel offsets from the first
// (+1 in x, +1 in y, +1 in z): the per-lane LDS address is then  base[class][row] + immediate, i.e. a
// dozen base registers instead of one address register per (pair, row).
//   pairs 0-8 : (kz,ky,kx=0) + (kz,ky,kx=1)      class 0, delta = 1
//   pairs 9-11: (kz,ky=0,kx=2) + (kz,ky=1,kx=2)  class 1, delta = HX
//   pair 12   : (kz=0,2,2) + (kz=1,2,2)          class 2, delta = HY*HX
//   pair 13   : (kz=2,2,2) + zero weights        class 0
__host__ __device__ constexpr int pair_first(int p) {
  return p < 9 ? (p / 3) * 9 + (p % 3) * 3 + 0 : (p < 12 ? (p - 9) * 9 + 0 * 3 + 2 : (p == 12 ? 0 * 9 + 2 * 3 + 2 : 26));
}
__host__ __device__ constexpr int pair_second(int p) {
  return p < 9 ? pair_first(p) + 1 : (p < 12 ? pair_first(p) + 3 : (p == 12 ? pair_first(p) + 9 : -1));
}
__host__ __device__ constexpr int pair_class(int p) { return p < 9 ? 0 : (p < 12 ? 1 : (p == 12 ? 2 : 0)); }
__host__ __device__ constexpr int tap_off(int tap) {     // voxel offset of a tap inside the halo tile
  return ((tap / 9) * HYs + (tap / 3) % 3) * HXs + tap % 3;
}

__global__ void __launch_bounds__(512, 2) conv3d_c16_f16x3_kernel(
    const float* __restrict__ x, const _Float16* __restrict__ wsplit, const float* __restrict__ bias,
    float* __restrict__ y, float* __restrict__ norm_out,
    int N, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z, int ntiles,
    float he, unsigned flags, float slope, float eps,
    const float* __restrict__ prev_y, const float* __restrict__ prev_norm, unsigned prev_flags,
    const float* __restrict__ amax_in, float* __restrict__ amax_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* raw = (float*)smem;                                         // fp32 halo (DMA target)
  _Float16* bhi = (_Float16*)(smem + RAW_FLOATS * 4);                // f16 hi plane [vox][16]
  _Float16* blo = bhi + HALF_ELEMS;                                  // f16 lo plane

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, kg = lane >> 4;
  if (tid < 16) ((unsigned*)(blo + HALF_ELEMS))[tid] = 0u;            // guard voxel (0 * garbage would be NaN)

  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t_begin = blockIdx.x * per;
  const int t_end = min(t_begin + per, ntiles);
  if (t_begin >= t_end) return;

  const long nvox = (long)D * H * W;
  const unsigned sample_bytes = (unsigned)(nvox * 64);

  // power-of-two input scale from the tensor's max-abs (gradient launches); 1 otherwise
  float in_scale = 1.f;
  if (amax_in != nullptr) {
    const float am = lf_amax_read(amax_in, lane);
    if (am > 0.f && am < 3.0e38f) {
      int ex;
      frexpf(am, &ex);                                               // am = m * 2^ex, m in [0.5, 1)
      in_scale = ldexpf(1.f, 13 - ex);                               // max maps into [2^12, 2^13)
    }
  }
  const float out_scale = he / in_scale;

  // ---- DMA piece constants (identical to the fp32 kernel) ----
  int rel[NITs], lxyz[NITs];
#pragma unroll
  for (int it = 0; it < NITs; ++it) {
    const int e = (wave + 8 * it) * 64 + lane;
    int v = e >> 2;
    const int q = e & 3;
    const int lx = v % HXs; v /= HXs;
    const int ly = v % HYs;
    const int lz = v / HYs;
    rel[it] = ((lz * H + ly) * W + lx) * 64 + q * 16;
    lxyz[it] = (e < HALOs * 4) ? (lx | (ly << 8) | (lz << 16)) : 0x7f7f7f;
  }

  // ---- weights (hi, lo) -> registers: [pair][term][cout 16][k 32] halfs ----
  // weights: hi and lo halves of all 14 pairs in registers (112 VGPRs)
  f16x8 whi[NPAIR], wlo[NPAIR];
  {
    const _Float16* wl = wsplit + li * 32 + kg * 8;
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      whi[p] = *(const f16x8*)(wl + (p * 2 + 0) * 512);
      wlo[p] = *(const f16x8*)(wl + (p * 2 + 1) * 512);
    }
  }
  f32x4 bv4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) bv4 = *(const f32x4*)(bias + kg * 4);

  // B-operand addressing: lanes kg 0,1 read the first tap of a pair (channels 0-7 / 8-15), lanes kg 2,3
  // the second tap.  rowvox[j] = halo voxel of (row j, x = li) at tap offset 0.
  // per-lane B base pointers (hi plane; the lo plane is a constant HALF_ELEMS further)
  const _Float16* bbase[3][4];
  {
    const int second = kg >> 1, chan8 = (kg & 1) * 8;
    const int delta[3] = {1, HXs, HYs * HXs};
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = wave * 4 + j;
        const int vox = ((r >> 3) * HYs + (r & 7)) * HXs + li + second * delta[c];
        bbase[c][j] = bhi + vox * 16 + chan8;
      }
  }

  auto issue_dma = [&](int t) {
    int tt = t;
    const int bx = tt % tiles_x; tt /= tiles_x;
    const int by = tt % tiles_y; tt /= tiles_y;
    const int bz = tt % tiles_z; tt /= tiles_z;
    const int ox = bx * TXs - 1, oy = by * TYs - 1, oz = bz * TZs - 1;
    const int tile_off = ((oz * H + oy) * W + ox) * 64;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long)tt * nvox * 16), 0, sample_bytes,
                                                                  0x00020000);
#pragma unroll
    for (int it = 0; it < NITs; ++it) {
      const int s = wave + 8 * it;
      if (s < NSLOTs) {
        const int gx = ox + (lxyz[it] & 0xff), gy = oy + ((lxyz[it] >> 8) & 0xff), gz = oz + (lxyz[it] >> 16);
        const bool ok = (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D;
        const int voff = ok ? (rel[it] + tile_off) : 0x7fffffff;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(raw + s * 256), 16, voff, 0, 0, 0);
      }
    }
  };

  // fp32 halo -> f16 hi / lo planes (each thread converts 9 float4 pieces)
  auto convert = [&]() {
#pragma unroll 1
    for (int it = 0; it < NITs; ++it) {
      const int e = tid + it * 512;                                   // float4 index, linear in the halo buffer
      if (e < HALOs * 4) {
        const f32x4 v = *(const f32x4*)(raw + e * 4) * in_scale;
        f16x4 h, l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          h[c] = (_Float16)v[c];
          l[c] = (_Float16)(v[c] - (float)h[c]);
        }
        *(f16x4*)(bhi + e * 4) = h;
        *(f16x4*)(blo + e * 4) = l;
      }
    }
  };

  f32x4 acc[4];
  auto compute = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (no explicit B double-buffer here: 112 VGPRs of weights leave no room; the partner wave on the
    //  SIMD covers the LDS latency)
#pragma unroll
    for (int p = 0; p < NPAIR; ++p) {
      constexpr int dummy = 0; (void)dummy;
      const int off = tap_off(pair_first(p)) * 16;                      // compile-time after unrolling
      const int cls = pair_class(p);
      f16x8 bh[4], bl[4];
      const f16x8 wl_p = wlo[p];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        bh[j] = *(const f16x8*)(bbase[cls][j] + off);
        bl[j] = *(const f16x8*)(bbase[cls][j] + off + HALF_ELEMS);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[p], bl[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl_p, bh[j], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(whi[p], bh[j], acc[j], 0, 0, 0);
    }
  };

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
    const int gx = bx * TXs + li;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = wave * 4 + j;
      const int gz = bz * TZs + (r >> 3), gy = by * TYs + (r & 7);
      const bool ok = gx < W && gy < H && gz < D;
      const int vox = ok ? (gz * H + gy) * W + gx : 0;
      pyv[j] = ok ? *(const f32x4*)(pybase + vox * 16 + kg * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      pnv[j] = (ok && pnbase) ? pnbase[vox] : 1.f;
    }
  };

  float wave_amax = 0.f;
  auto epilogue = [&](int t) {
    int tt = t;
    const int bx = tt % tiles_x; tt /= tiles_x;
    const int by = tt % tiles_y; tt /= tiles_y;
    const int bz = tt % tiles_z; tt /= tiles_z;
    float* ybase = y + (long)tt * nvox * 16;
    float* nbase = norm_out ? norm_out + (long)tt * nvox : nullptr;
    const int gx = bx * TXs + li;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = wave * 4 + j;
      const int gz = bz * TZs + (r >> 3), gy = by * TYs + (r & 7);
      const bool ok = gx < W && gy < H && gz < D;
      const int vox = ok ? (gz * H + gy) * W + gx : 0;
      if (prev_y != nullptr) {
        const f32x4 yp = pyv[j];
        f32x4 g = acc[j] * out_scale;
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
        if (ok) {
          *(f32x4*)(ybase + vox * 16 + kg * 4) = g;
          wave_amax = fmaxf(wave_amax, fmaxf(fmaxf(fabsf(g[0]), fabsf(g[1])), fmaxf(fabsf(g[2]), fabsf(g[3]))));
        }
        continue;
      }
      f32x4 v;
      float ss = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float u = acc[j][e] * out_scale + bv4[e];
        if (flags & LF_EPI_LRELU) u = fmaxf(u, u * slope);
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
      if (ok) {
        *(f32x4*)(ybase + vox * 16 + kg * 4) = v;
        if ((flags & LF_EPI_PIXELNORM) && nbase != nullptr && kg == 0) nbase[vox] = r_;
        wave_amax = fmaxf(wave_amax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      }
    }
  };

  // ---- pipeline: raw <- DMA(t+1) while MFMA(t) reads the f16 planes; convert between barriers ----
  issue_dma(t_begin);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  convert();
  __syncthreads();
  // Every wave stores tile t-1's results at the START of iteration t: the global stores then drain
  // behind tile t's MFMAs instead of stalling the vmcnt(0) that guards the barrier.
  for (int t = t_begin; t < t_end; ++t) {
    if (t + 1 < t_end) issue_dma(t + 1);
    if (t > t_begin) epilogue(t - 1);
    prefetch_prev(t);
    compute();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                  // everyone is done reading the f16 planes; DMA landed
    if (t + 1 < t_end) convert();
    __syncthreads();
  }
  epilogue(t_end - 1);
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
  // + one zeroed guard voxel: the zero-weight second tap of pair 13 reads one voxel past each plane
  const size_t shmem = (size_t)RAW_FLOATS * 4 + (size_t)HALF_ELEMS * 2 * 2 + 64;   // 69,632 + 69,120 + 64 B
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)conv3d_c16_f16x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)shmem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const unsigned grid = (unsigned)(pt < cus ? pt : cus);
  hipLaunchKernelGGL(conv3d_c16_f16x3_kernel, dim3(grid), dim3(512), shmem, (hipStream_t)stream, x, (const _Float16*)wsplit,
                     bias, y, norm_out, N, D, H, W, ptx, pty, ptz, (int)pt, he, flags, slope, eps, prev_y, prev_norm,
                     prev_flags, amax_in, amax_out);
  return lf_launch_status();
}
