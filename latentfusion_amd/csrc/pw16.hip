// Pointwise 16 -> 16 layers of the TRAINING step (1x1x1 convolutions under the bf16 autocast + storage policy: the sculptor's output
// block, reference recon/models.py:143,222 over modules/blocks.py:108-119 and modules/equalized.py) -- round 6.
//
// Until now such a layer ran on the 3x3x3 ring kernels with its weights on the centre tap (ops_train.pack_center_tap): 27 taps of
// MFMAs, halo staging and a z-plane ring for a product that touches one voxel -- 1.45 ms forward and 2.16 ms for the data gradient
// over 32 volumes of 128^3, plus a 0.83 ms lf_epilogue_bwd_c16 pass whose only job was the bias gradient.  A pointwise layer is one
// v_mfma_f32_16x16x16_bf16 per 16 voxels: lane (n = voxel, kg) loads ITS quarter of the voxel's record (the B operand: k = channels
// 4 kg .. +3, column n), holds W[row n][4 kg .. +3] as the A operand and ends with outputs 4 kg .. +3 of voxel n -- the quarter
// record it stores.  The kernels stream at the HBM: 32 B in + 32 B out per voxel.
//
// Arithmetic = the ring kernels' under the policy: products of bf16 values summed in fp32 on the matrix pipe, result rounded to
// bf16, `* he` rounded to bf16 (autocast's half-precision convolution and scaling), bias added in fp32, optional LeakyReLU, stored
// as bf16.  The data gradient is the same product with the transposed weights, rounded the same way; the bias gradient (sum of the
// incoming gradient over the voxels) rides along as per-workgroup partial sums reduced in a fixed order (deterministic).
#include "lf_common.h"

namespace {

typedef __bf16 bf16x4p __attribute__((ext_vector_type(4)));
typedef short s16x4p __attribute__((ext_vector_type(4)));

constexpr int PW_GPW = 8;                                         // groups of 16 voxels per wave and pass
constexpr int PW_BLOCKS = 2048;                                   // workgroups of the data gradient (each leaves one partial of the bias sums)

__device__ __forceinline__ float pw_rb16(float v) { return (float)(__bf16)v; }

__device__ __forceinline__ f32x4 pw_mfma(const bf16x4p a, const bf16x4p b) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4p, a), __builtin_bit_cast(s16x4p, b),
                                                   (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
}

template <bool IN16>
__device__ __forceinline__ bf16x4p pw_load(const void* __restrict__ x, long v, int kg) {
  if constexpr (IN16) return *(const bf16x4p*)((const __bf16*)x + v * 16 + kg * 4);
  else return __builtin_convertvector(*(const f32x4*)((const float*)x + v * 16 + kg * 4), bf16x4p);   // (what the ring kernels' staging does)
}

// y[v][co] = act( bf16(bf16(sum_ci W[co][ci] x[v][ci]) * he) + bias[co] )  as bf16
template <bool IN16, bool ACT>
__global__ void __launch_bounds__(256) pw16_fwd_kernel(const void* __restrict__ x, const __bf16* __restrict__ w, const float* __restrict__ bias,
                                                       float he, float slope, __bf16* __restrict__ y, long rows) {
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  const long groups = (rows + 15) >> 4;
  const bf16x4p a = *(const bf16x4p*)(w + n * 16 + kg * 4);      // W[co = n][ci = 4 kg .. +3]
  f32x4 b4 = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) b4 = *(const f32x4*)(bias + kg * 4);
  for (long g0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PW_GPW; g0 < groups; g0 += (long)gridDim.x * 4 * PW_GPW) {
    bf16x4p xb[PW_GPW];
#pragma unroll
    for (int gi = 0; gi < PW_GPW; ++gi) {
      const long v = min((g0 + gi) * 16 + n, rows - 1);            // (lanes past the end shadow the last voxel)
      xb[gi] = pw_load<IN16>(x, v, kg);
    }
#pragma unroll
    for (int gi = 0; gi < PW_GPW; ++gi) {
      const long v = (g0 + gi) * 16 + n;
      const f32x4 acc = pw_mfma(a, xb[gi]);
      f32x4 u;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = pw_rb16(pw_rb16(acc[e]) * he) + b4[e];
        if (ACT) t = fmaxf(t, t * slope);
        u[e] = t;
      }
      if (v < rows) *(bf16x4p*)(y + v * 16 + kg * 4) = __builtin_convertvector(u, bf16x4p);
    }
  }
}

// gx[v][ci] = bf16(bf16(sum_co Wt[ci][co] gy[v][co]) * he)   (stored bf16 or fp32);  partial[blockIdx][co] = sum over this
// workgroup's voxels of gy[v][co] (fp32, lanes in a fixed order)
template <bool OUT16>
__global__ void __launch_bounds__(256) pw16_bwd_kernel(const __bf16* __restrict__ gy, const __bf16* __restrict__ wt, float he,
                                                       void* __restrict__ gx, float* __restrict__ partial, long rows) {
  __shared__ f32x4 red[4][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 15, kg = lane >> 4;
  const long groups = (rows + 15) >> 4;
  const bf16x4p a = *(const bf16x4p*)(wt + n * 16 + kg * 4);     // Wt[ci = n][co = 4 kg .. +3]
  f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};                      // this lane's share of the bias gradient: channels 4 kg .. +3
  for (long g0 = ((long)blockIdx.x * 4 + wv) * PW_GPW; g0 < groups; g0 += (long)gridDim.x * 4 * PW_GPW) {
    bf16x4p gb[PW_GPW];
#pragma unroll
    for (int gi = 0; gi < PW_GPW; ++gi) {
      const long v = (g0 + gi) * 16 + n;
      gb[gi] = v < rows ? *(const bf16x4p*)(gy + v * 16 + kg * 4) : __builtin_convertvector((f32x4){0.f, 0.f, 0.f, 0.f}, bf16x4p);
    }
#pragma unroll
    for (int gi = 0; gi < PW_GPW; ++gi) {
      const long v = (g0 + gi) * 16 + n;
      const f32x4 acc = pw_mfma(a, gb[gi]);
      f32x4 u;
#pragma unroll
      for (int e = 0; e < 4; ++e) u[e] = pw_rb16(pw_rb16(acc[e]) * he);
      if (v < rows) {
        if constexpr (OUT16) *(bf16x4p*)((__bf16*)gx + v * 16 + kg * 4) = __builtin_convertvector(u, bf16x4p);
        else *(f32x4*)((float*)gx + v * 16 + kg * 4) = u;
      }
      if (partial != nullptr) bsum += __builtin_convertvector(gb[gi], f32x4);
    }
  }
  if (partial != nullptr) {
    // the 16 voxel lanes of a channel quarter (butterfly, fixed order), then the four waves in wave order
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) bsum[e] += __shfl_xor(bsum[e], o, 64);
    }
    if (n == 0) red[wv][kg] = bsum;
    __syncthreads();
    if (threadIdx.x < 4) {
      const int q = threadIdx.x;
      *(f32x4*)(partial + (long)blockIdx.x * 16 + q * 4) = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
    }
  }
}

// out[c] = sum over the workgroups' partials: lane l adds blocks l, l + 64, ... in order (fp64), then a fixed butterfly
__global__ void __launch_bounds__(64) pw16_bias_reduce_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out) {
  const int lane = threadIdx.x;
  for (int c = 0; c < 16; ++c) {
    double s = 0.0;
    for (int b = lane; b < nblk; b += 64) s += (double)partial[(long)b * 16 + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[c] = (float)s;
  }
}

inline unsigned pw_grid(long rows) {
  const long groups = (rows + 15) / 16;
  const long need = (groups + 4 * PW_GPW - 1) / (4 * PW_GPW);
  return (unsigned)(need < PW_BLOCKS ? need : PW_BLOCKS);
}

}  // namespace

extern "C" int lf_pw16_fwd(const void* x, int x_bf16, const void* w_bf16, const float* bias, float he, unsigned flags, float slope,
                           void* y, long rows, void* stream) {
  lf_clear_error();
  if (rows <= 0 || !x || !w_bf16 || !y || (flags & ~LF_EPI_LRELU)) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(y) || !lf_aligned16(w_bf16) || (bias && !lf_aligned16(bias))) return LF_EALIGN;
  const dim3 grid(pw_grid(rows)), block(256);
  hipStream_t s = (hipStream_t)stream;
  const bool act = (flags & LF_EPI_LRELU) != 0;
#define LAUNCH(I, A) hipLaunchKernelGGL((pw16_fwd_kernel<I, A>), grid, block, 0, s, x, (const __bf16*)w_bf16, bias, he, slope, (__bf16*)y, rows)
  if (x_bf16) { if (act) LAUNCH(true, true); else LAUNCH(true, false); }
  else        { if (act) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  return lf_launch_status();
}

extern "C" size_t lf_pw16_bwd_scratch_bytes(long rows) {
  lf_clear_error();
  if (rows <= 0) return 0;
  return (size_t)pw_grid(rows) * 16 * sizeof(float);
}

extern "C" int lf_pw16_bwd(const void* gy_bf16, const void* wt_bf16, float he, void* gx, int gx_bf16, float* gbias, void* scratch,
                           size_t scratch_bytes, long rows, void* stream) {
  lf_clear_error();
  if (rows <= 0 || !gy_bf16 || !wt_bf16 || !gx) return LF_EINVAL;
  if (gbias != nullptr && (scratch == nullptr || scratch_bytes < lf_pw16_bwd_scratch_bytes(rows))) return LF_ENOSPC;
  if (!lf_aligned16(gy_bf16) || !lf_aligned16(gx) || !lf_aligned16(wt_bf16) || (scratch && !lf_aligned16(scratch))) return LF_EALIGN;
  const dim3 grid(pw_grid(rows)), block(256);
  hipStream_t s = (hipStream_t)stream;
  float* partial = gbias != nullptr ? (float*)scratch : nullptr;
  if (gx_bf16) hipLaunchKernelGGL((pw16_bwd_kernel<true>), grid, block, 0, s, (const __bf16*)gy_bf16, (const __bf16*)wt_bf16, he, gx, partial, rows);
  else hipLaunchKernelGGL((pw16_bwd_kernel<false>), grid, block, 0, s, (const __bf16*)gy_bf16, (const __bf16*)wt_bf16, he, gx, partial, rows);
  if (gbias != nullptr) hipLaunchKernelGGL(pw16_bias_reduce_kernel, dim3(1), dim3(64), 0, s, partial, (int)grid.x, gbias);
  return lf_launch_status();
}
