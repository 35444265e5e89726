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

// The same on the fp32 matrix pipe (round 6): the lane-per-quarter form above reads every record four times through the L1 and
// takes 272 LDS-fed FMAs per voxel -- 0.59 ms per 8 x 128^3 launch, 0.47 of what its 2.2 GB cost at the HBM.  Here a wave takes 16
// voxels per step: lane (n = voxel, kg) loads ITS quarter record x[n][4 kg .. +3] once and contracts it as four
// v_mfma_f32_16x16x4_f32 steps -- step i pairs the lane's element i with the weight W[co = n][k = 4 kg + i], the K order is a
// permutation of the channels, the sums are the same -- and ends with outputs 4 kg .. +3 of voxel n: the quarter record it
// stores.  The depth input and the bias are one FMA each on the result; output 16 is a quarter dot product and two lane swaps.
constexpr int OIF_GPW = 8;                                        // groups of 16 voxels per wave
__global__ void __launch_bounds__(256) occ_input_fwd_mfma_kernel(const f32x4* __restrict__ z, const float* __restrict__ w,
                                                                 const float* __restrict__ b, f32x4* __restrict__ ta,
                                                                 float* __restrict__ t16, unsigned rows, unsigned D, unsigned P,
                                                                 float dstep, float slope) {
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  const unsigned g0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)OIF_GPW;   // first group of this wave
  const unsigned groups = (rows + 15u) >> 4;
  if (g0 >= groups) return;                                        // (wave-uniform)
  const f32x4 a4 = *(const f32x4*)(w + n * 20 + kg * 4);          // W[co = n][4 kg .. +3]
  const f32x4 w16 = *(const f32x4*)(w + 16 * 20 + kg * 4);        // W[16][4 kg .. +3]: this lane's share of output 16
  f32x4 wd, bb;                                                   // W[4 kg + j][16] (depth input) and the bias of outputs 4 kg .. +3
#pragma unroll
  for (int j = 0; j < 4; ++j) { wd[j] = w[(kg * 4 + j) * 20 + 16]; bb[j] = b[kg * 4 + j]; }
  const float wd16 = w[16 * 20 + 16], b16 = b[16];
  const unsigned ng = min((unsigned)OIF_GPW, groups - g0);
  f32x4 x[OIF_GPW];
#pragma unroll
  for (int gi = 0; gi < OIF_GPW; ++gi) {
    const unsigned v = (g0 + gi) * 16u + n;
    x[gi] = (gi < (int)ng && v < rows) ? z[(size_t)v * 4 + kg] : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // (P % 16 == 0: the 16 voxels of a group share their depth plane; one division per WAVE, then the plane index is stepped --
  //  with three integer divisions per lane and group these kernels were bound by their VALU, not by the HBM)
  unsigned pb = (g0 * 16u) % P, d = ((g0 * 16u) / P) % D;
#pragma unroll
  for (int gi = 0; gi < OIF_GPW; ++gi) {
    if (gi >= (int)ng) break;                                      // (wave-uniform)
    const unsigned v = (g0 + gi) * 16u + n;
    const bool live = v < rows;
    // torch.linspace(-1, 1, D): start + step * i in the first half, end - step * (D - 1 - i) in the second
    const float dc = D > 1 ? ((d < D / 2) ? -1.f + dstep * (float)d : 1.f - dstep * (float)(D - 1 - d)) : -1.f;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], x[gi][i], acc, 0, 0, 0);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sum = acc[j] + (bb[j] + wd[j] * dc);
      o[j] = sum > 0.f ? sum : sum * slope;
    }
    if (live) ta[(size_t)v * 4 + kg] = o;
    float s16 = w16[0] * x[gi][0] + w16[1] * x[gi][1] + w16[2] * x[gi][2] + w16[3] * x[gi][3];
    s16 += __shfl_xor(s16, 16, 64);
    s16 += __shfl_xor(s16, 32, 64);
    s16 += b16 + wd16 * dc;
    if (live && kg == 0) t16[v] = s16 > 0.f ? s16 : s16 * slope;
    pb += 16u;
    if (pb >= P) { pb = 0u; if (++d >= D) d = 0u; }
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
    const f32x4 g = gta[row * 4 + k];
    f32x4 t = (f32x4){1.f, 1.f, 1.f, 1.f};                          // (ta == NULL: gta already carries lrelu'(ta))
    if (ta != nullptr) t = ta[row * 4 + k];
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

// The input block's backward on the fp32 matrix pipe, with the factor projection's data gradient recomputed instead of read
// (round 6).  The lane-per-quarter form above is bound by its own loads and LDS-fed FMAs (0.96 ms per 8 x 128^3 launch; taking
// one of its five volume reads away moved it by 2 %).  Here lane (n = voxel, kg) loads ITS quarter of each record once:
//   W1[:16]^T gta   four v_mfma_f32_16x16x4_f32 steps, A = W1[j = 4 kg + i][c = n]  (gta already carries LeakyReLU'(ta))
//   g_zs            = (Wp_d^T gp2d[pixel]) * he_p for the voxel's depth d: four more steps, A = wpt[(d * 16 + n) * 16 + 4 kg + i],
//                   B = the lane's quarter of the 2-D gradient row (8 MB in all: L2) -- the 1 GB gradient volume the projection's
//                   data gradient used to write, and this kernel to read, does not exist; times wocc[voxel] it is the direct term
//   + W1[16][c] * gp16, then the epilogue backward of the layer that produced z, and the quarter record is stored.
__global__ void __launch_bounds__(256) occ_input_bwd_mfma_kernel(const f32x4* __restrict__ gta, const float* __restrict__ gp16,
                                                                 const float* __restrict__ w, const f32x4* __restrict__ gp2d,
                                                                 const float* __restrict__ wpt, float he_p,
                                                                 const float* __restrict__ wocc, f32x4* __restrict__ gz,
                                                                 unsigned rows, unsigned D, unsigned P, float slope,
                                                                 const f32x4* __restrict__ prev_y, const float* __restrict__ prev_norm,
                                                                 unsigned prev_flags) {
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  const unsigned g0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)OIF_GPW;
  const unsigned groups = (rows + 15u) >> 4;
  if (g0 >= groups) return;                                        // (wave-uniform)
  float a1[4];                                                    // W1[4 kg + i][n]
#pragma unroll
  for (int i = 0; i < 4; ++i) a1[i] = w[(kg * 4 + i) * 20 + n];
  const f32x4 w16 = *(const f32x4*)(w + 16 * 20 + kg * 4);        // W1[16][4 kg .. +3]
  const unsigned ng = min((unsigned)OIF_GPW, groups - g0);
  f32x4 xg[OIF_GPW], xp[OIF_GPW], xy[OIF_GPW], aw[OIF_GPW];
  float s16[OIF_GPW], so[OIF_GPW], sn[OIF_GPW];
  unsigned smp = (g0 * 16u) / (D * P), d = ((g0 * 16u) / P) % D, pb = (g0 * 16u) % P;   // (wave-uniform; stepped per group)
#pragma unroll
  for (int gi = 0; gi < OIF_GPW; ++gi) {
    const unsigned v = min((g0 + gi) * 16u + n, rows - 1u);        // (dead lanes shadow the last row)
    const bool on = gi < (int)ng;
    const unsigned pix = pb + n;
    xg[gi] = on ? gta[(size_t)v * 4 + kg] : (f32x4){0.f, 0.f, 0.f, 0.f};
    xp[gi] = on ? gp2d[((size_t)smp * P + pix) * 4 + kg] : (f32x4){0.f, 0.f, 0.f, 0.f};
    aw[gi] = on ? *(const f32x4*)(wpt + ((size_t)d * 16 + n) * 16 + kg * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    xy[gi] = (on && prev_y != nullptr) ? prev_y[(size_t)v * 4 + kg] : (f32x4){0.f, 0.f, 0.f, 0.f};
    s16[gi] = on ? gp16[v] : 0.f;
    so[gi] = on ? wocc[v] : 0.f;
    sn[gi] = (on && (prev_flags & LF_EPI_PIXELNORM)) ? prev_norm[v] : 1.f;
    pb += 16u;
    if (pb >= P) { pb = 0u; if (++d >= D) { d = 0u; ++smp; } }
  }
#pragma unroll
  for (int gi = 0; gi < OIF_GPW; ++gi) {
    if (gi >= (int)ng) break;                                      // (wave-uniform)
    const unsigned v = (g0 + gi) * 16u + n;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f}, gzs = acc;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[i], xg[gi][i], acc, 0, 0, 0);
      gzs = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[gi][i], xp[gi][i], gzs, 0, 0, 0);
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = ((gzs[j] * he_p) * so[gi] + acc[j]) + w16[j] * s16[gi];
    if (prev_y != nullptr) {
      const f32x4 vb = xy[gi];
      if (prev_flags & LF_EPI_PIXELNORM) {
        float dot = o[0] * vb[0] + o[1] * vb[1] + o[2] * vb[2] + o[3] * vb[3];
        dot += __shfl_xor(dot, 16, 64);
        dot += __shfl_xor(dot, 32, 64);
        dot *= (1.f / 16.f);
        const float r = sn[gi];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (o[e] - vb[e] * dot) / r;
      }
      if (prev_flags & LF_EPI_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = vb[e] > 0.f ? o[e] : o[e] * slope;
      }
    }
    if (v < rows) gz[(size_t)v * 4 + kg] = o;
  }
}

// Gradient of the occlusion weights through the factor projection (round 6): gw[v] = sum_c z[v][c] * g_zs[v][c] with
// g_zs = (Wp_d^T gp2d[pixel]) * he_p recomputed per 16 voxels as in the kernel above -- one read of z, 4 bytes out per voxel
// (was lf_conv1x1_bwd_data with LF_EPI_DOT: the generic pointwise kernel, 0.32 ms; before that lf_column_scale_bwd over two volumes).
__global__ void __launch_bounds__(256) occ_weight_grad_kernel(const f32x4* __restrict__ z, const f32x4* __restrict__ gp2d,
                                                              const float* __restrict__ wpt, float he_p, float* __restrict__ gw,
                                                              unsigned rows, unsigned D, unsigned P) {
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  const unsigned g0 = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (unsigned)OIF_GPW;
  const unsigned groups = (rows + 15u) >> 4;
  if (g0 >= groups) return;                                        // (wave-uniform)
  const unsigned ng = min((unsigned)OIF_GPW, groups - g0);
  f32x4 xz[OIF_GPW], xp[OIF_GPW], aw[OIF_GPW];
  unsigned smp = (g0 * 16u) / (D * P), d = ((g0 * 16u) / P) % D, pb = (g0 * 16u) % P;   // (wave-uniform; stepped per group)
#pragma unroll
  for (int gi = 0; gi < OIF_GPW; ++gi) {
    const unsigned v = min((g0 + gi) * 16u + n, rows - 1u);
    const bool on = gi < (int)ng;
    const unsigned pix = pb + n;
    xz[gi] = on ? z[(size_t)v * 4 + kg] : (f32x4){0.f, 0.f, 0.f, 0.f};
    xp[gi] = on ? gp2d[((size_t)smp * P + pix) * 4 + kg] : (f32x4){0.f, 0.f, 0.f, 0.f};
    aw[gi] = on ? *(const f32x4*)(wpt + ((size_t)d * 16 + n) * 16 + kg * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
    pb += 16u;
    if (pb >= P) { pb = 0u; if (++d >= D) { d = 0u; ++smp; } }
  }
#pragma unroll
  for (int gi = 0; gi < OIF_GPW; ++gi) {
    if (gi >= (int)ng) break;                                      // (wave-uniform)
    const unsigned v = (g0 + gi) * 16u + n;
    f32x4 gzs = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) gzs = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[gi][i], xp[gi][i], gzs, 0, 0, 0);
    const f32x4 zq = xz[gi];
    float dot = (gzs[0] * he_p) * zq[0] + (gzs[1] * he_p) * zq[1] + (gzs[2] * he_p) * zq[2] + (gzs[3] * he_p) * zq[3];
    dot += __shfl_xor(dot, 16, 64);
    dot += __shfl_xor(dot, 32, 64);
    if (kg == 0 && v < rows) gw[v] = dot;
  }
}

// The same followed by the depth softmax's backward in one kernel (round 6): a wave takes 16 pixels of one sample and walks ALL
// depths, so the column sum of the softmax gradient is a running sum in registers; the weights' gradients of the column wait in LDS
// (D x 16 floats per wave) for it:   glogits[d] = w[d] * (gw[d] - sum_e w[e] gw[e]).   No gw volume, no lf_column_softmax_bwd launch.
constexpr int OWS_CH = 8;                                         // depths per batch of loads
__global__ void __launch_bounds__(256) occ_weight_grad_softmax_kernel(const f32x4* __restrict__ z, const f32x4* __restrict__ gp2d,
                                                                      const float* __restrict__ wpt, float he_p,
                                                                      const float* __restrict__ wocc, float* __restrict__ glogits,
                                                                      unsigned D, unsigned P, unsigned ngroups) {
  extern __shared__ float sgw[];                                  // [4 waves][D][16]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, n = lane & 15, kg = lane >> 4;
  const unsigned grp = blockIdx.x * 4u + wv;                      // group of 16 pixels over all samples
  if (grp >= ngroups) return;                                      // (wave-uniform)
  const unsigned smp = grp / (P >> 4), pix = (grp % (P >> 4)) * 16u + n;
  float* mine = sgw + (size_t)wv * D * 16;
  const f32x4 xp = gp2d[((size_t)smp * P + pix) * 4 + kg];
  const size_t col = (size_t)smp * D * P + pix;                   // voxel (d = 0) of this lane's pixel
  float dotc = 0.f;
  for (unsigned d0 = 0; d0 < D; d0 += OWS_CH) {
    f32x4 xz[OWS_CH], aw[OWS_CH];
    float wd[OWS_CH];
#pragma unroll
    for (int i = 0; i < OWS_CH; ++i) {
      const unsigned d = min(d0 + i, D - 1u);
      xz[i] = z[(col + (size_t)d * P) * 4 + kg];
      aw[i] = *(const f32x4*)(wpt + ((size_t)d * 16 + n) * 16 + kg * 4);
      wd[i] = wocc[col + (size_t)d * P];
    }
#pragma unroll
    for (int i = 0; i < OWS_CH; ++i) {
      if (d0 + i >= D) break;                                      // (wave-uniform)
      f32x4 gzs = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 4; ++k) gzs = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[i][k], xp[k], gzs, 0, 0, 0);
      const f32x4 zq = xz[i];
      float dot = (gzs[0] * he_p) * zq[0] + (gzs[1] * he_p) * zq[1] + (gzs[2] * he_p) * zq[2] + (gzs[3] * he_p) * zq[3];
      dot += __shfl_xor(dot, 16, 64);
      dot += __shfl_xor(dot, 32, 64);
      if (kg == 0) mine[(d0 + i) * 16 + n] = dot;
      dotc += wd[i] * dot;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  for (unsigned d = kg; d < D; d += 4) glogits[col + (size_t)d * P] = wocc[col + (size_t)d * P] * (mine[d * 16 + n] - dotc);
}

// Data gradient of the 16 -> 1 output block onto the last 16-channel activation, with that layer's epilogue backward in the store
// (round 6; was lf_conv1x1_bwd_data with K = 1 padded to an MFMA step: 0.55 ms per 8 x 128^3 launch for 2.3 GB):
//   g[v][c] = lrelu'(y[v][c]) * (t[c] - y[v][c] * mean_c(t * y[v])) / norm[v],   t[c] = (gl[v] * w[c]) * he
// One lane per quarter record, the same products and order of the partial sums (the results differ from the pointwise kernel's in
// the last bit where the compiler contracts a multiply-add differently).
__global__ void __launch_bounds__(256) occ_head_bwd_kernel(const float* __restrict__ gl, const float* __restrict__ w16, float he,
                                                           const f32x4* __restrict__ y, const float* __restrict__ norm, unsigned flags,
                                                           float slope, f32x4* __restrict__ g, long rows) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int q = (int)(i & 3);
  const bool live = (i >> 2) < rows;
  const long row = live ? (i >> 2) : rows - 1;                    // (dead lanes shadow the last row: the shuffles stay defined)
  const f32x4 wq = *(const f32x4*)(w16 + q * 4);
  const f32x4 yp = y[row * 4 + q];
  const float gv = gl[row];
  f32x4 t;
#pragma unroll
  for (int e = 0; e < 4; ++e) t[e] = (gv * wq[e]) * he;
  if (flags & LF_EPI_PIXELNORM) {
    float dot = t[0] * yp[0] + t[1] * yp[1] + t[2] * yp[2] + t[3] * yp[3];
    dot += __shfl_xor(dot, 1, 64);
    dot += __shfl_xor(dot, 2, 64);
    dot *= (1.f / 16.f);
    const float rinv = 1.0f / norm[row];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (t[e] - yp[e] * dot) * rinv;
  }
  if (flags & LF_EPI_LRELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = yp[e] > 0.f ? t[e] : t[e] * slope;
  }
  if (live) g[row * 4 + q] = t;
}

// pre[v][co] = sum_taps w27[tap][co] * t16[v + tap - 1]  (zero padding), w27 = W2[:, 16] * he as [kz*9 + ky*3 + kx][16].
// A 16 x 27 by 27 x voxels product on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: M = the 16 outputs, N = 16 consecutive voxels,
// K = 28 tap slots in 7 steps, the last slot zero): lane (n = lane & 15, kg = lane >> 4) holds the A entry W[tap 4s + kg][co n]
// of every step for the whole kernel, loads ONE neighbour value per step as its B entry (voxel n, tap 4s + kg; unconditional
// load from a safe address, zero by select), and ends with outputs 4 kg .. 4 kg + 3 of voxel n -- a channels-last quarter record,
// so a wave's store covers 1 KB of consecutive bytes without a transpose.  (The VALU form -- 432 FMAs and 27 loads per voxel
// lane -- took 0.68 ms at 8 x 64^3; this one is bound by the 64 B per voxel it writes.)
constexpr int C17_GPW = 8;                                        // groups of 16 voxels per wave
__global__ void __launch_bounds__(256) occ_conv17_fwd_kernel(const float* __restrict__ t16, const float* __restrict__ w27,
                                                             f32x4* __restrict__ pre, int D, int H, int W, long rows, long groups) {
  const int lane = threadIdx.x & 63, n = lane & 15, kg = lane >> 4;
  const long g0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * C17_GPW;
  if (g0 >= groups) return;                                        // (wave-uniform)
  float a[7];
  int dz[7], dy[7], dx[7], dl[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int tap = 4 * s + kg;
    const bool tv = tap < 27;
    a[s] = tv ? w27[tap * 16 + n] : 0.f;
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    dz[s] = tv ? kz - 1 : 9999;                                    // (an impossible offset: slot 27 is never in range)
    dy[s] = ky - 1;
    dx[s] = kx - 1;
    dl[s] = ((kz - 1) * H + (ky - 1)) * W + (kx - 1);
  }
  const long v0 = g0 * 16 + n;
  int x = (int)(v0 % W), y = (int)((v0 / W) % H), z = (int)((v0 / ((long)W * H)) % D);
  // (round 6) all neighbour values of the wave's 8 groups are requested first (56 loads in flight per lane instead of 7 behind
  // each group's stores)
  float b[C17_GPW][7];
#pragma unroll
  for (int gi = 0; gi < C17_GPW; ++gi) {
    const long v = v0 + 16 * gi;
    const bool live = g0 + gi < groups && v < rows;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const bool ok = live && (unsigned)(z + dz[s]) < (unsigned)D && (unsigned)(y + dy[s]) < (unsigned)H && (unsigned)(x + dx[s]) < (unsigned)W;
      b[gi][s] = 0.f;
      if (ok) b[gi][s] = t16[v + dl[s]];
    }
    x += 16;
    while (x >= W) {
      x -= W;
      if (++y >= H) {
        y = 0;
        if (++z >= D) z = 0;
      }
    }
  }
#pragma unroll
  for (int gi = 0; gi < C17_GPW; ++gi) {
    if (g0 + gi >= groups) break;                                  // (wave-uniform)
    const long v = v0 + 16 * gi;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 7; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[gi][s], acc, 0, 0, 0);
    if (v < rows) pre[v * 4 + kg] = acc;
  }
}

// gp16[u] = lrelu'(t16[u]) * sum_taps sum_co w27[tap][co] * g[u - (tap - 1)][co], in two stages per z plane:
//   A  h[tap][v] = sum_co w27[tap][co] g[v][co] for the 18 x 18 voxels around a 16 x 16 footprint, on the fp32 matrix pipe
//      (v_mfma_f32_16x16x4_f32: M = 32 tap slots in two blocks, N = 16 voxels, K = the 16 channels; a lane's B entries are the
//      four floats of the quarter record it loads), written to LDS as [tap][voxel];
//   B  every thread owns one (y, x) of the footprint and adds the nine h values of each kz that land on it to the running sums
//      of the output planes p - 1, p, p + 1; a plane is complete -- and stored, times lrelu'(t16) -- once the plane behind it
//      has been added.
// A workgroup walks C17_ZR output planes down z, so every gradient record is loaded (18 / 16)^2 (1 + 2 / C17_ZR) = 1.4 times
// instead of 13.5 times by the lane-per-quarter form it replaces (0.89 ms at 8 x 64^3, bound by those loads).
constexpr int C17_ZR = 16, C17_HS = 336;                         // planes per workgroup; LDS slots per tap (324 halo voxels, padded to 21 x 16)
constexpr int C17_LS = 340;                                       // LDS row stride per tap: 4 rows apart = 16 banks apart (stage A's four lane groups write two-way, not four-way)
__global__ void __launch_bounds__(256) occ_conv17_bwd_kernel(const f32x4* __restrict__ g, const float* __restrict__ t16,
                                                             const float* __restrict__ w27, float* __restrict__ gp16,
                                                             int D, int H, int W, int nzr, int nty, int ntx, float slope) {
  __shared__ float hT[28 * C17_LS];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = lane & 15, kg = lane >> 4;
  int b = blockIdx.x;
  const int bx = b % ntx; b /= ntx;
  const int by = b % nty; b /= nty;
  const int bz = b % nzr;
  const long smp = b / nzr;
  const int x0 = bx * 16, y0 = by * 16, z0 = bz * C17_ZR;
  // A operands: block blk, step s: W[tap = 16 blk + n][co = 4 kg + s]
  float a[2][4];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int tap = 16 * blk + n;
      a[blk][s] = tap < 27 ? w27[tap * 16 + 4 * kg + s] : 0.f;
    }
  const int ty = tid >> 4, tx = tid & 15;                          // stage B: this thread's output column
  const bool col_ok = y0 + ty < H && x0 + tx < W;
  const long vol = (long)D * H * W;
  float accm = 0.f, acc0 = 0.f, accp = 0.f;                        // sums of output planes p - 1, p, p + 1
  // (round 6) a wave's <= 6 record groups of a plane are requested together, and the NEXT plane's before this plane's barrier and
  // stage B: the kernel ran one dependent load per group and wave at a time (0.51 ms per 8 x 128^3 launch = 2.2 TB/s of the
  // gradient volume it reads)
  constexpr int NGW = (C17_HS / 16 + 3) / 4;                       // groups per wave: 6 (waves 0: 0, 4, .., 20)
  f32x4 rg[NGW];
  auto load_plane = [&](int p) {
    const bool pin = p >= 0 && p < D;
#pragma unroll
    for (int k = 0; k < NGW; ++k) {
      const int grp = wv + 4 * k;
      const int q = grp * 16 + n;
      const int yy = y0 - 1 + q / 18, xx = x0 - 1 + q % 18;
      const bool ok = pin && grp < C17_HS / 16 && q < 324 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
      rg[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (ok) rg[k] = g[(smp * vol + ((long)p * H + yy) * W + xx) * 4 + kg];
    }
  };
  load_plane(z0 - 1);
  for (int p = z0 - 1; p <= z0 + C17_ZR && p <= D; ++p) {
    const bool plane_in = p >= 0 && p < D;                         // (workgroup-uniform)
    if (plane_in) {
      // ---- stage A: 21 groups of 16 halo voxels, wave wv takes groups wv, wv + 4, ...
#pragma unroll
      for (int k = 0; k < NGW; ++k) {
        const int grp = wv + 4 * k;
        if (grp >= C17_HS / 16) break;                             // (wave-uniform)
        const int q = grp * 16 + n;                                // halo slot of this lane's voxel
        const f32x4 r = rg[k];
        f32x4 d0 = (f32x4){0.f, 0.f, 0.f, 0.f}, d1 = d0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][s], r[s], d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1][s], r[s], d1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          hT[(4 * kg + i) * C17_LS + q] = d0[i];                   // taps 0..15
          if (16 + 4 * kg + i < 28) hT[(16 + 4 * kg + i) * C17_LS + q] = d1[i];   // taps 16..27 (27 = the zero slot)
        }
      }
    }
    if (p + 1 <= z0 + C17_ZR && p + 1 <= D) load_plane(p + 1);     // in flight across the barrier and stage B
    __syncthreads();
    if (plane_in) {
      // ---- stage B: source voxel of tap (kz, ky, kx) for output (ty, tx) = halo slot (ty + 2 - ky, tx + 2 - kx) of this plane
      float sm = 0.f, s0 = 0.f, sp = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int slot = (ty + 2 - ky) * 18 + (tx + 2 - kx);
          sm += hT[(0 * 9 + ky * 3 + kx) * C17_LS + slot];
          s0 += hT[(1 * 9 + ky * 3 + kx) * C17_LS + slot];
          sp += hT[(2 * 9 + ky * 3 + kx) * C17_LS + slot];
        }
      accm += sm;
      acc0 += s0;
      accp += sp;
    }
    // output plane p - 1 has now received planes p - 2, p - 1, p
    const int zo = p - 1;
    if (col_ok && zo >= z0 && zo < z0 + C17_ZR && zo < D) {
      const long u = smp * vol + ((long)zo * H + (y0 + ty)) * W + (x0 + tx);
      gp16[u] = t16[u] > 0.f ? accm : accm * slope;
    }
    accm = acc0;
    acc0 = accp;
    accp = 0.f;
    __syncthreads();
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
  if (rows < 0x7fffffffL && P < 0x7fffffffL && (P & 15) == 0 && lf_aligned16(w)) {
    const long groups = (rows + 15) / 16;
    hipLaunchKernelGGL(occ_input_fwd_mfma_kernel, dim3((unsigned)((groups + 4 * OIF_GPW - 1) / (4 * OIF_GPW))), dim3(256), 0, (hipStream_t)stream,
                       (const f32x4*)z, w, b, (f32x4*)ta, t16, (unsigned)rows, (unsigned)D, (unsigned)P, dstep, slope);
    return lf_launch_status();
  }
  hipLaunchKernelGGL(occ_input_fwd_kernel, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)z, w, b,
                     (f32x4*)ta, t16, rows, D, P, dstep, slope);
  return lf_launch_status();
}

extern "C" int lf_occ_input_bwd(const float* gta, const float* ta, const float* gp16, const float* w, const float* g_zs,
                                const float* wocc, float* gz, long rows, float slope, const float* prev_y, const float* prev_norm,
                                unsigned prev_flags, void* stream) {
  lf_clear_error();
  if (rows <= 0 || !gta || !gp16 || !w || !gz || ((g_zs == nullptr) != (wocc == nullptr))) return LF_EINVAL;
  if ((prev_flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) || (prev_y == nullptr && prev_flags != 0) ||
      ((prev_flags & LF_EPI_PIXELNORM) && prev_norm == nullptr)) return LF_EINVAL;
  if (!lf_aligned16(gta) || (ta && !lf_aligned16(ta)) || !lf_aligned16(gz) || (g_zs && !lf_aligned16(g_zs)) ||
      (prev_y && !lf_aligned16(prev_y))) return LF_EALIGN;
  hipLaunchKernelGGL(occ_input_bwd_kernel, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const f32x4*)gta,
                     (const f32x4*)ta, gp16, w, (const f32x4*)g_zs, wocc, (f32x4*)gz, rows, slope, (const f32x4*)prev_y, prev_norm,
                     prev_flags);
  return lf_launch_status();
}

extern "C" int lf_occ_input_bwd_proj(const float* gta, const float* gp16, const float* w, const float* gp2d, const float* wpack_t,
                                     float he_p, const float* wocc, float* gz, int N, int D, long P, float slope,
                                     const float* prev_y, const float* prev_norm, unsigned prev_flags, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || !gta || !gp16 || !w || !gp2d || !wpack_t || !wocc || !gz) return LF_EINVAL;
  if ((prev_flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) || (prev_y == nullptr && prev_flags != 0) ||
      ((prev_flags & LF_EPI_PIXELNORM) && prev_norm == nullptr)) return LF_EINVAL;
  const long rows = (long)N * D * P;
  if (rows >= 0x7fffffffL || (P & 15)) return LF_EINVAL;             // (a group of 16 voxels lies in one depth plane)
  if (!lf_aligned16(gta) || !lf_aligned16(gz) || !lf_aligned16(gp2d) || !lf_aligned16(wpack_t) || !lf_aligned16(w) ||
      (prev_y && !lf_aligned16(prev_y))) return LF_EALIGN;
  const long groups = (rows + 15) / 16;
  hipLaunchKernelGGL(occ_input_bwd_mfma_kernel, dim3((unsigned)((groups + 4 * OIF_GPW - 1) / (4 * OIF_GPW))), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)gta, gp16, w, (const f32x4*)gp2d, wpack_t, he_p, wocc, (f32x4*)gz, (unsigned)rows, (unsigned)D,
                     (unsigned)P, slope, (const f32x4*)prev_y, prev_norm, prev_flags);
  return lf_launch_status();
}

extern "C" int lf_occ_weight_grad(const float* z, const float* gp2d, const float* wpack_t, float he_p, float* gw, int N, int D, long P,
                                  void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || !z || !gp2d || !wpack_t || !gw) return LF_EINVAL;
  const long rows = (long)N * D * P;
  if (rows >= 0x7fffffffL || (P & 15)) return LF_EINVAL;
  if (!lf_aligned16(z) || !lf_aligned16(gp2d) || !lf_aligned16(wpack_t)) return LF_EALIGN;
  const long groups = (rows + 15) / 16;
  hipLaunchKernelGGL(occ_weight_grad_kernel, dim3((unsigned)((groups + 4 * OIF_GPW - 1) / (4 * OIF_GPW))), dim3(256), 0, (hipStream_t)stream,
                     (const f32x4*)z, (const f32x4*)gp2d, wpack_t, he_p, gw, (unsigned)rows, (unsigned)D, (unsigned)P);
  return lf_launch_status();
}

extern "C" int lf_occ_weight_grad_softmax_bwd(const float* z, const float* gp2d, const float* wpack_t, float he_p, const float* weights,
                                              float* glogits, int N, int D, long P, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || P <= 0 || !z || !gp2d || !wpack_t || !weights || !glogits) return LF_EINVAL;
  const long rows = (long)N * D * P;
  if (rows >= 0x7fffffffL || (P & 15) || D > 256) return LF_EINVAL;    // (D <= 256: the columns of four waves fit the default 64 KB of LDS)
  if (!lf_aligned16(z) || !lf_aligned16(gp2d) || !lf_aligned16(wpack_t)) return LF_EALIGN;
  const long ngroups = (long)N * (P / 16);
  const size_t lds = (size_t)4 * D * 16 * sizeof(float);
  hipLaunchKernelGGL(occ_weight_grad_softmax_kernel, dim3((unsigned)((ngroups + 3) / 4)), dim3(256), lds, (hipStream_t)stream,
                     (const f32x4*)z, (const f32x4*)gp2d, wpack_t, he_p, weights, glogits, (unsigned)D, (unsigned)P, (unsigned)ngroups);
  return lf_launch_status();
}

extern "C" int lf_occ_head_bwd(const float* gl, const float* w16, float he, const float* y, const float* norm, unsigned flags,
                               float slope, float* g, long rows, void* stream) {
  lf_clear_error();
  if (rows <= 0 || !gl || !w16 || !y || !g || (flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) ||
      ((flags & LF_EPI_PIXELNORM) && norm == nullptr)) return LF_EINVAL;
  if (!lf_aligned16(y) || !lf_aligned16(g) || !lf_aligned16(w16)) return LF_EALIGN;
  hipLaunchKernelGGL(occ_head_bwd_kernel, dim3((unsigned)((rows * 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, gl, w16, he,
                     (const f32x4*)y, norm, flags, slope, (f32x4*)g, rows);
  return lf_launch_status();
}

extern "C" int lf_occ_conv17_fwd(const float* t16, const float* w27, float* pre, int N, int D, int H, int W, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || !t16 || !w27 || !pre) return LF_EINVAL;
  if (!lf_aligned16(pre)) return LF_EALIGN;
  const long rows = (long)N * D * H * W;
  const long groups = (rows + 15) / 16;
  hipLaunchKernelGGL(occ_conv17_fwd_kernel, dim3((unsigned)((groups + 4 * C17_GPW - 1) / (4 * C17_GPW))), dim3(256), 0, (hipStream_t)stream, t16,
                     w27, (f32x4*)pre, D, H, W, rows, groups);
  return lf_launch_status();
}

extern "C" int lf_occ_conv17_bwd(const float* g, const float* t16, const float* w27, float* gp16, int N, int D, int H, int W,
                                 float slope, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || !g || !t16 || !w27 || !gp16) return LF_EINVAL;
  if (!lf_aligned16(g)) return LF_EALIGN;
  const int nzr = (D + C17_ZR - 1) / C17_ZR, nty = (H + 15) / 16, ntx = (W + 15) / 16;
  const long blocks = (long)N * nzr * nty * ntx;
  if (blocks >= 0x7fffffffL) return LF_EINVAL;
  hipLaunchKernelGGL(occ_conv17_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4*)g, t16, w27, gp16, D, H, W,
                     nzr, nty, ntx, slope);
  return lf_launch_status();
}
