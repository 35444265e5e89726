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
#include "lf_common.h"

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

// gw[tap][co][ci] = scale * sum over blocks (fixed order, fp64)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                           int nblk, int taps, int ntiles, int ncit, int Cin, int Cout,
                                                           float scale) {
  const int tap = blockIdx.x, tile = blockIdx.y, t = threadIdx.x;
  double s = 0.0;
  for (int b = 0; b < nblk; ++b) s += (double)partial[(((long)b * taps + tap) * ntiles + tile) * 256 + t];
  const int ct = tile / ncit, cit = tile - ct * ncit;
  const int co = ct * 16 + (t >> 4), ci = cit * 16 + (t & 15);
  if (co < Cout && ci < Cin) gw[((long)tap * Cout + co) * Cin + ci] = (float)(s * (double)scale);
}

struct WgradPlan { int taps, nct, ncit, chunk, nblk; };

bool wgrad_plan(int dims, long total, int Cin, int Cout, bool ones, WgradPlan& p) {
  if (dims != 0 && dims != 2 && dims != 3) return false;
  if (total <= 0 || total >= 0x7fffffffL || Cout <= 0 || (!ones && Cin <= 0)) return false;
  p.taps = dims == 3 ? 27 : (dims == 2 ? 9 : 1);
  p.nct = (Cout + 15) / 16;
  p.ncit = ones ? 1 : (Cin + 15) / 16;
  // enough blocks to fill the chip for small problems, at most ~2048 partials per output for big ones
  long chunk = 1024;
  while ((total + chunk - 1) / chunk > 2048) chunk *= 2;
  p.chunk = (int)chunk;
  p.nblk = (int)((total + chunk - 1) / chunk);
  return (long)p.nct * p.ncit <= 65535;
}

}  // namespace

extern "C" size_t lf_conv_bwd_weight_scratch_bytes(int dims, int N, int D, int H, int W, int Cin, int Cout) {
  WgradPlan p;
  if (!wgrad_plan(dims, (long)N * D * H * W, Cin > 0 ? Cin : 1, Cout, Cin <= 0, p)) return 0;
  return (size_t)p.nblk * p.taps * p.nct * p.ncit * 256 * sizeof(float);
}

extern "C" int lf_conv_bwd_weight(const float* x, const float* gpre, float* gw, void* scratch, size_t scratch_bytes,
                                  int dims, int N, int D, int H, int W, int Cin, int Cout, float scale, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || gpre == nullptr || gw == nullptr) return LF_EINVAL;
  const bool ones = (x == nullptr);
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
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(p.taps, p.nct * p.ncit), block, 0, s, partial, gw, p.nblk, p.taps,
                     p.nct * p.ncit, p.ncit, Cin, Cout, scale);
  return lf_launch_status();
}
