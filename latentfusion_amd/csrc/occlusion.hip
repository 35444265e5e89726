// Stages of the occlusion module of the renderer that handle its 17th channel, sequenced explicitly by the render-loop engine
// (round 5):
//   Photographer.forward, latentfusion/recon/models.py:378-395,427-430: occ = UNet3d(cat(z, depth coordinate)) -> softmax over D
//   UNet3d input block, latentfusion/modules/unet.py + blocks.py:78-91: 1x1x1 conv (C+1 -> C+1) * he + bias, LeakyReLU
//   first convolution of the U-Net, blocks.py:152-158: 3x3x3, C+1 -> C
// The 17-channel tensors of that module (16 latent channels + the normalised voxel depth) never exist here: the input block
// reads the 16-channel volume, takes the depth coordinate from the voxel's plane index, and writes its outputs as a 16-channel
// channels-last volume (outputs 0..15) plus a SCALAR volume (output 16).  The 17 -> 16 convolution behind it is then the
// 16 -> 16 Winograd kernel over the former (lf_conv3d_c16_wino, LF_EPI_ADD form) with the 1 -> 16 convolution of the scalar
// volume (lf_occ_conv17_fwd, HBM-bound: 4 B read + 64 B written per voxel) as its addend; backward mirrors it.
#include "lf_common.h"

namespace {

// w [17][20] = W1 * he (row j = output channel: 16 latent inputs, then the depth input at [16], 3 pad), b [20] (17 used).
// One lane per (voxel, output quarter q): the weights come from LDS (the row index depends on the lane).
__global__ void __launch_bounds__(256) occ_input_fwd_kernel(const f32x4* __restrict__ z, const float* __restrict__ w,
                                                            const float* __restrict__ b, f32x4* __restrict__ ta, float* __restrict__ t16,
                                                            long rows, int D, long P, float dstep, float slope) {
  __shared__ __attribute__((aligned(16))) float sw[17 * 20 + 20];
  for (int i = threadIdx.x; i < 17 * 20 + 20; i += 256) sw[i] = i < 340 ? w[i] : b[i - 340];
  __syncthreads();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = i >> 2;
  const int q = (int)(i & 3);
  if (row >= rows) return;
  const int d = (int)((row / P) % D);
  // torch.linspace(-1, 1, D): start + step * i in the first half, end - step * (D - 1 - i) in the second
  const float dc = D > 1 ? ((d < D / 2) ? -1.f + dstep * (float)d : 1.f - dstep * (float)(D - 1 - d)) : -1.f;
  f32x4 x[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = z[row * 4 + k];              // (the four lanes of a voxel read the same 64-byte line)
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float* wr = sw + (q * 4 + e) * 20;
    float s = sw[340 + q * 4 + e] + wr[16] * dc;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const f32x4 wv = *(const f32x4*)(wr + k * 4);
#pragma unroll
      for (int c = 0; c < 4; ++c) s += wv[c] * x[k][c];
    }
    o[e] = s > 0.f ? s : s * slope;
  }
  ta[row * 4 + q] = o;
  // output 16: each lane of the voxel takes a quarter of the inputs, fixed-order butterfly over the four lanes
  {
    const f32x4 wv = *(const f32x4*)(sw + 16 * 20 + q * 4);
    float s = wv[0] * x[q][0] + wv[1] * x[q][1] + wv[2] * x[q][2] + wv[3] * x[q][3];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += sw[340 + 16] + sw[16 * 20 + 16] * dc;
    if (q == 0) t16[row] = s > 0.f ? s : s * slope;
  }
}

// gz[c] = g_zs[c] * wocc + sum_{j<16} w[j][c] * gta_j * lrelu'(ta_j) + w[16][c] * gp16   (gp16 already carries lrelu'(t16));
// with prev_y: the epilogue backward of the layer that produced z (its saved output prev_y, norm prev_norm) is applied to gz
// before the store, as the convolution data-gradient kernels do (lf_conv3x3_bwd_data)
__global__ void __launch_bounds__(256) occ_input_bwd_kernel(const f32x4* __restrict__ gta, const f32x4* __restrict__ ta,
                                                            const float* __restrict__ gp16, const float* __restrict__ w,
                                                            const f32x4* __restrict__ g_zs, const float* __restrict__ wocc,
                                                            f32x4* __restrict__ gz, long rows, float slope,
                                                            const f32x4* __restrict__ prev_y, const float* __restrict__ prev_norm,
                                                            unsigned prev_flags) {
  __shared__ __attribute__((aligned(16))) float sw[17 * 20];
  for (int i = threadIdx.x; i < 17 * 20; i += 256) sw[i] = w[i];
  __syncthreads();
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int q = (int)(i & 3);
  const bool live = (i >> 2) < rows;
  const long row = live ? (i >> 2) : rows - 1;                    // (dead lanes shadow the last row: the shuffles stay defined)
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (g_zs != nullptr) acc = g_zs[row * 4 + q] * wocc[row];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x4 g = gta[row * 4 + k], t = ta[row * 4 + k];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float gp = t[c] > 0.f ? g[c] : g[c] * slope;
      acc += *(const f32x4*)(sw + (k * 4 + c) * 20 + q * 4) * gp;
    }
  }
  acc += *(const f32x4*)(sw + 16 * 20 + q * 4) * gp16[row];
  if (prev_y != nullptr) {
    const f32x4 vb = prev_y[row * 4 + q];
    if (prev_flags & LF_EPI_PIXELNORM) {
      float dot = acc[0] * vb[0] + acc[1] * vb[1] + acc[2] * vb[2] + acc[3] * vb[3];
      dot += __shfl_xor(dot, 1, 64);
      dot += __shfl_xor(dot, 2, 64);
      dot *= (1.f / 16.f);
      const float r = prev_norm[row];
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = (acc[e] - vb[e] * dot) / r;
    }
    if (prev_flags & LF_EPI_LRELU) {
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[e] = vb[e] > 0.f ? acc[e] : acc[e] * slope;
    }
  }
  if (live) gz[row * 4 + q] = acc;
}

// pre[v][co] = sum_taps w27[tap][co] * t16[v + tap - 1]  (zero padding), w27 = W2[:, 16] * he as [kz*9 + ky*3 + kx][16].
// One lane per voxel, all 16 outputs (the weights are wave-uniform: scalar loads); the loads are unconditional on clamped
// addresses (no branch separates them: a lane has its 27 loads in flight together) and the records go out through an LDS
// transpose so that each store instruction of a wave covers 1 KB of consecutive bytes.
__global__ void __launch_bounds__(256) occ_conv17_fwd_kernel(const float* __restrict__ t16, const float* __restrict__ w27,
                                                             f32x4* __restrict__ pre, int N, int D, int H, int W) {
  __shared__ __attribute__((aligned(16))) float st[4][64 * 20];
  const long rows = (long)N * D * H * W;
  const long row = (long)blockIdx.x * 256 + threadIdx.x;
  const bool live = row < rows;
  const long rc = live ? row : rows - 1;
  const int x = (int)(rc % W), y = (int)((rc / W) % H), zc = (int)((rc / ((long)W * H)) % D);
  const long plane0 = rc - ((long)zc * H + y) * W - x;             // first voxel of this sample
  float acc[16];
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
  for (int kz = 0; kz < 3; ++kz) {
    const int zz = zc + kz - 1;
    const int zq = min(max(zz, 0), D - 1);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      const int yq = min(max(yy, 0), H - 1);
      const bool okzy = zz >= 0 && zz < D && yy >= 0 && yy < H;
      const long line = plane0 + ((long)zq * H + yq) * W;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        const float v = t16[line + min(max(xx, 0), W - 1)];
        const float t = (okzy && xx >= 0 && xx < W) ? v : 0.f;
        const float* wr = w27 + ((kz * 3 + ky) * 3 + kx) * 16;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] += wr[c] * t;
      }
    }
  }
  const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < 4; ++k) *(f32x4*)(&st[wv][l * 20 + k * 4]) = (f32x4){acc[k * 4], acc[k * 4 + 1], acc[k * 4 + 2], acc[k * 4 + 3]};
  __syncthreads();
  const long row0 = (long)blockIdx.x * 256 + wv * 64;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int v = (l >> 2) + 16 * k;
    if (row0 + v < rows) pre[(row0 + v) * 4 + (l & 3)] = *(const f32x4*)(&st[wv][v * 20 + (l & 3) * 4]);
  }
}

// gp16[u] = lrelu'(t16[u]) * sum_taps sum_co w27[tap][co] * g[u - (tap - 1)][co]: a lane takes FOUR voxels along W and a
// quarter of the channels (each loaded record serves up to three outputs), then a fixed-order butterfly over the four lanes.
// Branch-free like the forward: clamped addresses, out-of-range records replaced by zeros.
__global__ void __launch_bounds__(256) occ_conv17_bwd_kernel(const f32x4* __restrict__ g, const float* __restrict__ t16,
                                                             const float* __restrict__ w27, float* __restrict__ gp16,
                                                             int N, int D, int H, int W, int W4, float slope) {
  __shared__ __attribute__((aligned(16))) float sw[27 * 16];
  for (int i = threadIdx.x; i < 27 * 16; i += 256) sw[i] = w27[i];
  __syncthreads();
  const long groups = (long)N * D * H * W4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int q = (int)(i & 3);
  const bool live = (i >> 2) < groups;
  const long grp = live ? (i >> 2) : groups - 1;
  const int x0 = (int)(grp % W4) * 4, y = (int)((grp / W4) % H), zc = (int)((grp / ((long)W4 * H)) % D);
  const long n = grp / ((long)W4 * H * D);
  float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1                                                   // (a real loop: 18 records in flight per lane, not 54)
  for (int kz = 0; kz < 3; ++kz) {
    const int zz = zc - (kz - 1);
    const int zq = min(max(zz, 0), D - 1);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y - (ky - 1);
      const int yq = min(max(yy, 0), H - 1);
      const bool okzy = zz >= 0 && zz < D && yy >= 0 && yy < H;
      const long line = ((n * D + zq) * H + yq) * (long)W;
      f32x4 wv[3];
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) wv[kx] = *(const f32x4*)(sw + ((kz * 3 + ky) * 3 + kx) * 16 + q * 4);
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const int xp = x0 + j - 1;                               // position of the loaded record
        f32x4 r = g[(line + min(max(xp, 0), W - 1)) * 4 + q];
        if (!(okzy && xp >= 0 && xp < W)) r = (f32x4){0.f, 0.f, 0.f, 0.f};           // (a select, not a branch around the load)
#pragma unroll
        for (int oi = 0; oi < 4; ++oi) {
          const int kx = oi - j + 2;                             // xp = (x0 + oi) - (kx - 1)
          if (kx < 0 || kx > 2) continue;
          o[oi] += wv[kx][0] * r[0] + wv[kx][1] * r[1] + wv[kx][2] * r[2] + wv[kx][3] * r[3];
        }
      }
    }
  }
#pragma unroll
  for (int oi = 0; oi < 4; ++oi) {
    o[oi] += __shfl_xor(o[oi], 1, 64);
    o[oi] += __shfl_xor(o[oi], 2, 64);
  }
  const float mine = q == 0 ? o[0] : q == 1 ? o[1] : q == 2 ? o[2] : o[3];
  const int xo = x0 + q;
  if (live && xo < W) {
    const long u = ((n * D + zc) * H + y) * (long)W + xo;
    gp16[u] = t16[u] > 0.f ? mine : mine * slope;
  }
}

}  // namespace

extern "C" int lf_occ_input_fwd(const float* z, const float* w, const float* b, float* ta, float* t16, int N, int D, long P,
                                float slope, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || !z || !w || !b || !ta || !t16) return LF_EINVAL;
  if (!lf_aligned16(z) || !lf_aligned16(ta)) return LF_EALIGN;
  const long rows = (long)N * D * P;
  const float dstep = D > 1 ? 2.f / (float)(D - 1) : 0.f;
  hipLaunchKernelGGL(occ_input_fwd_kernel, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)z, w, b,
                     (f32x4*)ta, t16, rows, D, P, dstep, slope);
  return lf_launch_status();
}

extern "C" int lf_occ_input_bwd(const float* gta, const float* ta, const float* gp16, const float* w, const float* g_zs,
                                const float* wocc, float* gz, long rows, float slope, const float* prev_y, const float* prev_norm,
                                unsigned prev_flags, void* stream) {
  lf_clear_error();
  if (rows <= 0 || !gta || !ta || !gp16 || !w || !gz || ((g_zs == nullptr) != (wocc == nullptr))) return LF_EINVAL;
  if ((prev_flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) || (prev_y == nullptr && prev_flags != 0) ||
      ((prev_flags & LF_EPI_PIXELNORM) && prev_norm == nullptr)) return LF_EINVAL;
  if (!lf_aligned16(gta) || !lf_aligned16(ta) || !lf_aligned16(gz) || (g_zs && !lf_aligned16(g_zs)) ||
      (prev_y && !lf_aligned16(prev_y))) return LF_EALIGN;
  hipLaunchKernelGGL(occ_input_bwd_kernel, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)gta,
                     (const f32x4*)ta, gp16, w, (const f32x4*)g_zs, wocc, (f32x4*)gz, rows, slope, (const f32x4*)prev_y, prev_norm,
                     prev_flags);
  return lf_launch_status();
}

extern "C" int lf_occ_conv17_fwd(const float* t16, const float* w27, float* pre, int N, int D, int H, int W, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || !t16 || !w27 || !pre) return LF_EINVAL;
  if (!lf_aligned16(pre)) return LF_EALIGN;
  const long rows = (long)N * D * H * W;
  hipLaunchKernelGGL(occ_conv17_fwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, t16, w27, (f32x4*)pre,
                     N, D, H, W);
  return lf_launch_status();
}

extern "C" int lf_occ_conv17_bwd(const float* g, const float* t16, const float* w27, float* gp16, int N, int D, int H, int W,
                                 float slope, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || !g || !t16 || !w27 || !gp16) return LF_EINVAL;
  if (!lf_aligned16(g)) return LF_EALIGN;
  const int W4 = (W + 3) / 4;
  const long groups = (long)N * D * H * W4;
  hipLaunchKernelGGL(occ_conv17_bwd_kernel, dim3((unsigned)((groups * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g, t16,
                     w27, gp16, N, D, H, W, W4, slope);
  return lf_launch_status();
}
