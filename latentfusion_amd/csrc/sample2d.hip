// 2-D grid sampling of planar (NCHW) images: the F.grid_sample(align_corners=False) calls of
//   crop_to_viewport / Camera.zoom      latentfusion/modules/geometry.py:20-44,287-354   (zeros padding)
//   Camera.uncrop                        latentfusion/modules/geometry.py:261-285         (border padding)
//   ibr.reproject_views                  latentfusion/ibr.py:52-93                        (zeros padding)
// grid: [N][Ho][Wo][2] = (x, y) in [-1, 1]; modes bilinear / nearest (round-half-even like ATen).
// One thread per output pixel, looping over the (few) channels; backward w.r.t. the grid is a plain
// per-pixel sum, backward w.r.t. the image scatters with float atomics (not used on any ranking path).
#include "lf_common.h"

namespace {

__device__ __forceinline__ float unnorm(float g, int size) { return ((g + 1.f) * (float)size - 1.f) * 0.5f; }

// border padding: clip to [0, size-1]; returns d(clipped)/d(p)
__device__ __forceinline__ float clip_border(float& p, int size) {
  const float hi = (float)(size - 1);
  const float d = (p > 0.f && p < hi) ? 1.f : 0.f;
  p = fminf(fmaxf(p, 0.f), hi);
  return d;
}

template <bool BILINEAR, bool BORDER>
__global__ void __launch_bounds__(256) sample2d_fwd_kernel(const float* __restrict__ img, const float* __restrict__ grid,
                                                           float* __restrict__ out, int N, int C, int H, int W, int Ho,
                                                           int Wo) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * Ho * Wo) return;
  const int n = idx / (Ho * Wo), pix = idx - n * Ho * Wo;
  float px = unnorm(grid[(long)idx * 2 + 0], W), py = unnorm(grid[(long)idx * 2 + 1], H);
  if (BORDER) { clip_border(px, W); clip_border(py, H); }
  const float* src = img + (long)n * C * H * W;
  float* dst = out + (long)n * C * Ho * Wo + pix;
  if (BILINEAR) {
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = px - fx, ty = py - fy;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
    const float w00 = (1.f - tx) * (1.f - ty), w01 = tx * (1.f - ty), w10 = (1.f - tx) * ty, w11 = tx * ty;
    for (int c = 0; c < C; ++c) {
      const float* s = src + (long)c * H * W;
      float v = 0.f;
      if (vy0 && vx0) v += s[y0 * W + x0] * w00;
      if (vy0 && vx1) v += s[y0 * W + x1] * w01;
      if (vy1 && vx0) v += s[y1 * W + x0] * w10;
      if (vy1 && vx1) v += s[y1 * W + x1] * w11;
      dst[(long)c * Ho * Wo] = v;
    }
  } else {
    const int xn = (int)nearbyintf(px), yn = (int)nearbyintf(py);
    const bool ok = (unsigned)xn < (unsigned)W && (unsigned)yn < (unsigned)H;
    for (int c = 0; c < C; ++c) dst[(long)c * Ho * Wo] = ok ? src[(long)c * H * W + yn * W + xn] : 0.f;
  }
}

template <bool BILINEAR, bool BORDER>
__global__ void __launch_bounds__(256) sample2d_bwd_kernel(const float* __restrict__ img, const float* __restrict__ grid,
                                                           const float* __restrict__ gout, float* __restrict__ gimg,
                                                           float* __restrict__ ggrid, int N, int C, int H, int W, int Ho,
                                                           int Wo) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * Ho * Wo) return;
  const int n = idx / (Ho * Wo), pix = idx - n * Ho * Wo;
  float px = unnorm(grid[(long)idx * 2 + 0], W), py = unnorm(grid[(long)idx * 2 + 1], H);
  float mx = 0.5f * (float)W, my = 0.5f * (float)H;            // d(p)/d(grid)
  if (BORDER) { mx *= clip_border(px, W); my *= clip_border(py, H); }
  const float* src = img + (long)n * C * H * W;
  float* gsrc = gimg ? gimg + (long)n * C * H * W : nullptr;
  const float* go = gout + (long)n * C * Ho * Wo + pix;
  float gx = 0.f, gy = 0.f;
  if (BILINEAR) {
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float tx = px - fx, ty = py - fy;
    const bool vx0 = (unsigned)x0 < (unsigned)W, vx1 = (unsigned)x1 < (unsigned)W;
    const bool vy0 = (unsigned)y0 < (unsigned)H, vy1 = (unsigned)y1 < (unsigned)H;
    for (int c = 0; c < C; ++c) {
      const float g = go[(long)c * Ho * Wo];
      const float* s = src + (long)c * H * W;
      const float v00 = (vy0 && vx0) ? s[y0 * W + x0] : 0.f, v01 = (vy0 && vx1) ? s[y0 * W + x1] : 0.f;
      const float v10 = (vy1 && vx0) ? s[y1 * W + x0] : 0.f, v11 = (vy1 && vx1) ? s[y1 * W + x1] : 0.f;
      gx += g * ((v01 - v00) * (1.f - ty) + (v11 - v10) * ty);
      gy += g * ((v10 - v00) * (1.f - tx) + (v11 - v01) * tx);
      if (gsrc) {
        float* d = gsrc + (long)c * H * W;
        if (vy0 && vx0) atomicAdd(d + y0 * W + x0, g * (1.f - tx) * (1.f - ty));
        if (vy0 && vx1) atomicAdd(d + y0 * W + x1, g * tx * (1.f - ty));
        if (vy1 && vx0) atomicAdd(d + y1 * W + x0, g * (1.f - tx) * ty);
        if (vy1 && vx1) atomicAdd(d + y1 * W + x1, g * tx * ty);
      }
    }
  } else if (gsrc) {
    const int xn = (int)nearbyintf(px), yn = (int)nearbyintf(py);
    if ((unsigned)xn < (unsigned)W && (unsigned)yn < (unsigned)H)
      for (int c = 0; c < C; ++c) atomicAdd(gsrc + (long)c * H * W + yn * W + xn, go[(long)c * Ho * Wo]);
  }
  if (ggrid) {
    ggrid[(long)idx * 2 + 0] = gx * mx;
    ggrid[(long)idx * 2 + 1] = gy * my;
  }
}

}  // namespace

extern "C" int lf_grid_sample2d_fwd(const float* img, const float* grid, float* out, int N, int C, int H, int W, int Ho,
                                    int Wo, int bilinear, int border, void* stream) {
  lf_clear_error();
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return LF_EINVAL;
  if ((long)N * Ho * Wo >= 0x7fffffffL || (long)H * W >= 0x7fffffffL) return LF_EINVAL;
  dim3 g((unsigned)(((long)N * Ho * Wo + 255) / 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(B, P) hipLaunchKernelGGL((sample2d_fwd_kernel<B, P>), g, b, 0, s, img, grid, out, N, C, H, W, Ho, Wo)
  if (bilinear) { if (border) LAUNCH(true, true); else LAUNCH(true, false); }
  else          { if (border) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  return lf_launch_status();
}

extern "C" int lf_grid_sample2d_bwd(const float* img, const float* grid, const float* gout, float* gimg, float* ggrid,
                                    int N, int C, int H, int W, int Ho, int Wo, int bilinear, int border, void* stream) {
  lf_clear_error();
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return LF_EINVAL;
  if ((long)N * Ho * Wo >= 0x7fffffffL || (long)H * W >= 0x7fffffffL) return LF_EINVAL;
  dim3 g((unsigned)(((long)N * Ho * Wo + 255) / 256)), b(256);
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(B, P) hipLaunchKernelGGL((sample2d_bwd_kernel<B, P>), g, b, 0, s, img, grid, gout, gimg, ggrid, N, C, H, W, Ho, Wo)
  if (bilinear) { if (border) LAUNCH(true, true); else LAUNCH(true, false); }
  else          { if (border) LAUNCH(false, true); else LAUNCH(false, false); }
#undef LAUNCH
  return lf_launch_status();
}
