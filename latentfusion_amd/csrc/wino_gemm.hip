// Winograd F(2x2x2, 3x3x3) for WIDE 3-D convolutions (e.g. the released model's 256 -> 256 blocks on 16^3
// volumes, latentfusion/modules/blocks.py:152-158 with tools/train/train.sh:37-43): with hundreds of channels
// the transforms amortise, so the three classic stages are separate launches and the 64 per-frequency
// [tiles x Cin] x [Cin x Cout] products go to the library GEMM (rocBLAS through torch.bmm, fp32 MFMA):
//   lf_wino3d_input_transform :  x [N][D][H][W][Cin]  ->  V [64][T][Cin]        (T = N * ceil(D/2) ceil(H/2) ceil(W/2))
//   (host) M[f] = V[f] @ U[f]                         ->  M [64][T][Cout]
//   lf_wino3d_output_transform:  M -> y [N][D][H][W][Cout]  with He scale, bias, LeakyReLU, PixelNorm fused
// Both kernels put the channel axis on the lanes (float4 per lane), so every global access is a contiguous
// row of channels.  All arithmetic fp32.
#include "lf_common.h"

namespace {

// one wave per (tile, z-frequency a); lanes loop over channel quads
__global__ void __launch_bounds__(256) wino3d_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int D,
                                                          int H, int W, int C, int tz, int ty, int tx, long T) {
  const int lane = threadIdx.x & 63, a = threadIdx.x >> 6;
  const long tile = blockIdx.x;
  long r = tile;
  const int bx = (int)(r % tx); r /= tx;
  const int by = (int)(r % ty); r /= ty;
  const int bz = (int)(r % tz);
  const int n = (int)(r / tz);
  const int z0 = 2 * bz - 1, y0 = 2 * by - 1, x0 = 2 * bx - 1;
  // z input transform of frequency a: d0-d2, d1+d2, d2-d1, d1-d3
  const int dza = (a == 0) ? 0 : (a == 2 ? 2 : 1);
  const int dzb = (a == 0) ? 2 : (a == 1 ? 2 : (a == 2 ? 1 : 3));
  const float sb = (a == 1) ? 1.f : -1.f;
  const int za = z0 + dza, zb = z0 + dzb;
  const bool za_ok = (unsigned)za < (unsigned)D, zb_ok = (unsigned)zb < (unsigned)D;
  const float* xs = x + (long)n * D * H * W * C;
  for (int q = lane; q * 4 < C; q += 64) {
    f32x4 vx[4][4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      const int yy = y0 + dy;
      const bool y_ok = (unsigned)yy < (unsigned)H;
      f32x4 d[4];
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int xx = x0 + dx;
        const bool ok = y_ok && (unsigned)xx < (unsigned)W;
        f32x4 va = (f32x4){0.f, 0.f, 0.f, 0.f}, vb = va;
        if (ok && za_ok) va = *(const f32x4*)(xs + (((long)za * H + yy) * W + xx) * C + q * 4);
        if (ok && zb_ok) vb = *(const f32x4*)(xs + (((long)zb * H + yy) * W + xx) * C + q * 4);
        d[dx] = va + vb * sb;
      }
      vx[dy][0] = d[0] - d[2];
      vx[dy][1] = d[1] + d[2];
      vx[dy][2] = d[2] - d[1];
      vx[dy][3] = d[1] - d[3];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 v = (b == 0) ? (vx[0][c] - vx[2][c]) : (b == 1) ? (vx[1][c] + vx[2][c])
                      : (b == 2) ? (vx[2][c] - vx[1][c]) : (vx[1][c] - vx[3][c]);
        *(f32x4*)(V + ((long)(a * 16 + b * 4 + c) * T + tile) * C + q * 4) = v;
      }
  }
}

// one wave per tile: lanes = output-channel quads (Cout <= 256 for the fused PixelNorm)
__global__ void __launch_bounds__(256) wino3d_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                           float* __restrict__ y, float* __restrict__ norm_out, int N, int D,
                                                           int H, int W, int C, int tz, int ty, int tx, long T, float he,
                                                           unsigned flags, float slope, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tile = (long)blockIdx.x * 4 + wave;
  if (tile >= T) return;
  long r = tile;
  const int bx = (int)(r % tx); r /= tx;
  const int by = (int)(r % ty); r /= ty;
  const int bz = (int)(r % tz);
  const int n = (int)(r / tz);
  const bool pn = (flags & LF_EPI_PIXELNORM) != 0;
  float ss[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) ss[k] = 0.f;
  // every lane runs every pass (idle channel quads carry zeros) so that the wave reduction below is convergent
  const int passes = (C / 4 + 63) / 64;
  for (int ps = 0; ps < passes; ++ps) {                        // (single pass when C <= 256)
    const int q = ps * 64 + lane;
    const bool live = q * 4 < C;
    f32x4 o[2][4];                                              // [z][j*2+i]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      f32x4 Yj[4];                                              // y/x output transform of z-frequency a
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        f32x4 m[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          m[c] = live ? *(const f32x4*)(M + ((long)(a * 16 + b * 4 + c) * T + tile) * C + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 t0 = m[0] + m[1] + m[2], t1 = m[1] - m[2] - m[3];
        if (b == 0) { Yj[0] = t0; Yj[1] = t1; }
        if (b == 1) { Yj[0] += t0; Yj[1] += t1; Yj[2] = t0; Yj[3] = t1; }
        if (b == 2) { Yj[0] += t0; Yj[1] += t1; Yj[2] -= t0; Yj[3] -= t1; }
        if (b == 3) { Yj[2] -= t0; Yj[3] -= t1; }
      }
#pragma unroll
      for (int ji = 0; ji < 4; ++ji) {
        if (a == 0) { o[0][ji] = Yj[ji]; }
        if (a == 1) { o[0][ji] += Yj[ji]; o[1][ji] = Yj[ji]; }
        if (a == 2) { o[0][ji] += Yj[ji]; o[1][ji] -= Yj[ji]; }
        if (a == 3) { o[1][ji] -= Yj[ji]; }
      }
    }
    const f32x4 bv = (bias && live) ? *(const f32x4*)(bias + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int zo = 0; zo < 2; ++zo)
#pragma unroll
      for (int ji = 0; ji < 4; ++ji) {
        f32x4 v = o[zo][ji] * he + bv;
        if (flags & LF_EPI_LRELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
        }
        o[zo][ji] = v;
        ss[zo * 4 + ji] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
      }
    if (!pn || C <= 256) {
      // C <= 256: this is the only pass, the wave holds every channel of the 8 voxels
      float rinv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        rinv[k] = 1.f;
        if (pn) {
          const float tq = lf_wave_sum(ss[k]) / (float)C + eps;
          rinv[k] = 1.f / sqrtf(tq);
          ss[k] = sqrtf(tq);
        }
      }
#pragma unroll
      for (int zo = 0; zo < 2; ++zo)
#pragma unroll
        for (int ji = 0; ji < 4; ++ji) {
          const int gz = 2 * bz + zo, gy = 2 * by + (ji >> 1), gx = 2 * bx + (ji & 1);
          if (live && gz < D && gy < H && gx < W) {
            const long vox = (((long)n * D + gz) * H + gy) * W + gx;
            *(f32x4*)(y + vox * C + q * 4) = o[zo][ji] * rinv[zo * 4 + ji];
            if (pn && norm_out != nullptr && lane == 0) norm_out[vox] = ss[zo * 4 + ji];
          }
        }
    } else {
      // wider than a wave: store un-normalised, the caller runs lf_pixelnorm_fwd
#pragma unroll
      for (int zo = 0; zo < 2; ++zo)
#pragma unroll
        for (int ji = 0; ji < 4; ++ji) {
          const int gz = 2 * bz + zo, gy = 2 * by + (ji >> 1), gx = 2 * bx + (ji & 1);
          if (live && gz < D && gy < H && gx < W)
            *(f32x4*)(y + ((((long)n * D + gz) * H + gy) * W + gx) * C + q * 4) = o[zo][ji];
        }
    }
  }
}


// ---- 2-D: F(2x2, 3x3), 16 frequencies, tiles of 2 x 2 pixels (the released model's 2-D U-Nets) ----------------
// one wave per tile; lanes loop over channel quads
__global__ void __launch_bounds__(256) wino2d_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int H,
                                                          int W, int C, int ty, int tx, long T) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tile = (long)blockIdx.x * 4 + wave;
  if (tile >= T) return;
  long r = tile;
  const int bx = (int)(r % tx); r /= tx;
  const int by = (int)(r % ty);
  const int n = (int)(r / ty);
  const int y0 = 2 * by - 1, x0 = 2 * bx - 1;
  const float* xs = x + (long)n * H * W * C;
  for (int q = lane; q * 4 < C; q += 64) {
    f32x4 vx[4][4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
      const int yy = y0 + dy;
      f32x4 d[4];
#pragma unroll
      for (int dx = 0; dx < 4; ++dx) {
        const int xx = x0 + dx;
        d[dx] = ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) ? *(const f32x4*)(xs + ((long)yy * W + xx) * C + q * 4)
                                                                         : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      vx[dy][0] = d[0] - d[2];
      vx[dy][1] = d[1] + d[2];
      vx[dy][2] = d[2] - d[1];
      vx[dy][3] = d[1] - d[3];
    }
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4 v = (b == 0) ? (vx[0][c] - vx[2][c]) : (b == 1) ? (vx[1][c] + vx[2][c])
                      : (b == 2) ? (vx[2][c] - vx[1][c]) : (vx[1][c] - vx[3][c]);
        *(f32x4*)(V + ((long)(b * 4 + c) * T + tile) * C + q * 4) = v;
      }
  }
}

__global__ void __launch_bounds__(256) wino2d_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                           float* __restrict__ y, float* __restrict__ norm_out, int N, int H,
                                                           int W, int C, int ty, int tx, long T, float he, unsigned flags,
                                                           float slope, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long tile = (long)blockIdx.x * 4 + wave;
  if (tile >= T) return;
  long r = tile;
  const int bx = (int)(r % tx); r /= tx;
  const int by = (int)(r % ty);
  const int n = (int)(r / ty);
  const bool pn = (flags & LF_EPI_PIXELNORM) != 0;
  float ss[4] = {0.f, 0.f, 0.f, 0.f};
  const int passes = (C / 4 + 63) / 64;
  for (int ps = 0; ps < passes; ++ps) {
    const int q = ps * 64 + lane;
    const bool live = q * 4 < C;
    f32x4 o[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      f32x4 m[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
        m[c] = live ? *(const f32x4*)(M + ((long)(b * 4 + c) * T + tile) * C + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
      const f32x4 t0 = m[0] + m[1] + m[2], t1 = m[1] - m[2] - m[3];
      if (b == 0) { o[0] = t0; o[1] = t1; }
      if (b == 1) { o[0] += t0; o[1] += t1; o[2] = t0; o[3] = t1; }
      if (b == 2) { o[0] += t0; o[1] += t1; o[2] -= t0; o[3] -= t1; }
      if (b == 3) { o[2] -= t0; o[3] -= t1; }
    }
    const f32x4 bv = (bias && live) ? *(const f32x4*)(bias + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ji = 0; ji < 4; ++ji) {
      f32x4 v = o[ji] * he + bv;
      if (flags & LF_EPI_LRELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], v[e] * slope);
      }
      o[ji] = v;
      ss[ji] += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const bool fuse = !pn || C <= 256;
    float rinv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      rinv[k] = 1.f;
      if (pn && fuse) {
        const float tq = lf_wave_sum(ss[k]) / (float)C + eps;
        rinv[k] = 1.f / sqrtf(tq);
        ss[k] = sqrtf(tq);
      }
    }
#pragma unroll
    for (int ji = 0; ji < 4; ++ji) {
      const int gy = 2 * by + (ji >> 1), gx = 2 * bx + (ji & 1);
      if (live && gy < H && gx < W) {
        const long pix = ((long)n * H + gy) * W + gx;
        *(f32x4*)(y + pix * C + q * 4) = o[ji] * rinv[ji];
        if (pn && fuse && norm_out != nullptr && lane == 0) norm_out[pix] = ss[ji];
      }
    }
  }
}

}  // namespace

extern "C" long lf_wino3d_tiles(int N, int D, int H, int W) {
  return (long)N * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2);
}

extern "C" int lf_wino3d_input_transform(const float* x, float* V, int N, int D, int H, int W, int C, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(V)) return LF_EALIGN;
  const long T = lf_wino3d_tiles(N, D, H, W);
  if (T >= 0x7fffffffL) return LF_EINVAL;
  hipLaunchKernelGGL(wino3d_input_kernel, dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, x, V, N, D, H, W, C,
                     (D + 1) / 2, (H + 1) / 2, (W + 1) / 2, T);
  return lf_launch_status();
}

// flags: LF_EPI_*; with LF_EPI_PIXELNORM and C > 256 the output is left un-normalised (norm_out untouched)
extern "C" int lf_wino3d_output_transform(const float* M, const float* bias, float* y, float* norm_out, int N, int D, int H,
                                          int W, int C, float he, unsigned flags, float slope, float eps, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return LF_EINVAL;
  if (!lf_aligned16(M) || !lf_aligned16(y) || (bias && !lf_aligned16(bias))) return LF_EALIGN;
  const long T = lf_wino3d_tiles(N, D, H, W);
  if (T >= 0x7fffffffL) return LF_EINVAL;
  hipLaunchKernelGGL(wino3d_output_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, M, bias, y,
                     norm_out, N, D, H, W, C, (D + 1) / 2, (H + 1) / 2, (W + 1) / 2, T, he, flags, slope, eps);
  return lf_launch_status();
}

extern "C" long lf_wino2d_tiles(int N, int H, int W) { return (long)N * ((H + 1) / 2) * ((W + 1) / 2); }

extern "C" int lf_wino2d_input_transform(const float* x, float* V, int N, int H, int W, int C, void* stream) {
  lf_clear_error();
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(V)) return LF_EALIGN;
  const long T = lf_wino2d_tiles(N, H, W);
  if (T >= 0x7fffffffL) return LF_EINVAL;
  hipLaunchKernelGGL(wino2d_input_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, V, N, H, W, C,
                     (H + 1) / 2, (W + 1) / 2, T);
  return lf_launch_status();
}

extern "C" int lf_wino2d_output_transform(const float* M, const float* bias, float* y, float* norm_out, int N, int H, int W,
                                          int C, float he, unsigned flags, float slope, float eps, void* stream) {
  lf_clear_error();
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return LF_EINVAL;
  if (!lf_aligned16(M) || !lf_aligned16(y) || (bias && !lf_aligned16(bias))) return LF_EALIGN;
  const long T = lf_wino2d_tiles(N, H, W);
  if (T >= 0x7fffffffL) return LF_EINVAL;
  hipLaunchKernelGGL(wino2d_output_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, (hipStream_t)stream, M, bias, y,
                     norm_out, N, H, W, C, (H + 1) / 2, (W + 1) / 2, T, he, flags, slope, eps);
  return lf_launch_status();
}
