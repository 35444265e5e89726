// Fixed-factor x2 / x0.5 rescaling at the end of a Block (Interpolate, latentfusion/modules/
// __init__.py:18-36 = F.interpolate(scale_factor, mode in {nearest, bilinear, trilinear},
// align_corners=False)), channels-last, forward and adjoint.  The adjoint is a gather over the
// (at most 4 per axis) output positions that reference an input position: deterministic, no atomics.
#include "lf_common.h"

namespace {

// 1-D source taps of output index d: value = (1-w1)*in[i0] + w1*in[i1]
__device__ __forceinline__ void axis_taps(int d, int n_in, int linear, int up, int& i0, int& i1, float& w1) {
  if (!linear) {
    i0 = up ? (d >> 1) : (d << 1);                         // floor(d * (1/scale))
    i0 = min(i0, n_in - 1);
    i1 = i0; w1 = 0.f;
    return;
  }
  float src = up ? ((float)d + 0.5f) * 0.5f - 0.5f : ((float)d + 0.5f) * 2.f - 0.5f;
  src = fmaxf(src, 0.f);                                   // area_pixel_compute_source_index clamps at 0
  i0 = min((int)floorf(src), n_in - 1);
  i1 = min(i0 + 1, n_in - 1);
  w1 = src - (float)i0;
}

__global__ void __launch_bounds__(256) resize_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                                                         int vec, int linear, int up, int dims) {
  const int lpv = vec ? C / 4 : C;
  const long per = (long)Do * Ho * Wo * lpv;
  const int n = blockIdx.y;
  const float* xs = x + (long)n * Di * Hi * Wi * C;
  float* ys = y + (long)n * Do * Ho * Wo * C;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += (long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % lpv);
    long v = idx / lpv;
    const int ox = (int)(v % Wo); v /= Wo;
    const int oy = (int)(v % Ho);
    const int oz = (int)(v / Ho);
    int x0, x1, y0, y1, z0 = 0, z1 = 0;
    float wx, wy, wz = 0.f;
    axis_taps(ox, Wi, linear, up, x0, x1, wx);
    axis_taps(oy, Hi, linear, up, y0, y1, wy);
    if (dims == 3) axis_taps(oz, Di, linear, up, z0, z1, wz);
    const long sW = C, sH = (long)Wi * C, sD = (long)Hi * Wi * C;
    const long o = (((long)oz * Ho + oy) * Wo + ox) * C;
    const int ne = vec ? 4 : 1;
    for (int e = 0; e < ne; ++e) {
      const long c = vec ? q * 4 + e : q;
      const float* b = xs + c;
      const float v00 = b[z0 * sD + y0 * sH + x0 * sW] * (1.f - wx) + b[z0 * sD + y0 * sH + x1 * sW] * wx;
      const float v01 = b[z0 * sD + y1 * sH + x0 * sW] * (1.f - wx) + b[z0 * sD + y1 * sH + x1 * sW] * wx;
      float r = v00 * (1.f - wy) + v01 * wy;
      if (dims == 3 && linear) {
        const float v10 = b[z1 * sD + y0 * sH + x0 * sW] * (1.f - wx) + b[z1 * sD + y0 * sH + x1 * sW] * wx;
        const float v11 = b[z1 * sD + y1 * sH + x0 * sW] * (1.f - wx) + b[z1 * sD + y1 * sH + x1 * sW] * wx;
        r = r * (1.f - wz) + (v10 * (1.f - wy) + v11 * wy) * wz;
      }
      ys[o + c] = r;
    }
  }
}

// nearest x2 / x0.5, C % 4 == 0: one thread per 16 bytes, pure index mapping (the released model's rescaling mode; the
// generic kernel above spends its time on 64-bit div / mod and four scalar taps per element)
__global__ void __launch_bounds__(256) resize_nearest_vec4_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y,
                                                                  int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C4, int up) {
  const unsigned per = (unsigned)Do * Ho * Wo * C4;
  const int n = blockIdx.y;
  const f32x4* xs = x + (long)n * Di * Hi * Wi * C4;
  f32x4* ys = y + (long)n * per;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < per; idx += gridDim.x * 256u) {
    const unsigned q = idx % (unsigned)C4;
    unsigned v = idx / (unsigned)C4;
    const unsigned ox = v % (unsigned)Wo; v /= (unsigned)Wo;
    const unsigned oy = v % (unsigned)Ho;
    const unsigned oz = v / (unsigned)Ho;
    const unsigned sx = min(up ? ox >> 1 : ox << 1, (unsigned)Wi - 1u), sy = min(up ? oy >> 1 : oy << 1, (unsigned)Hi - 1u),
                   sz = min(up ? oz >> 1 : oz << 1, (unsigned)Di - 1u);
    ys[idx] = xs[((sz * (unsigned)Hi + sy) * (unsigned)Wi + sx) * (unsigned)C4 + q];
  }
}

// (bi|tri)linear x2 / x0.5, C % 4 == 0: one thread per 16 bytes, 32-bit index math, float4 taps (the U-Nets' rescaling mode:
// create_blocks' default, latentfusion/modules/unet.py:24-27 -> blocks.py:10); same tap formula as the generic kernel
template <int DIMS>
__global__ void __launch_bounds__(256) resize_linear_vec4_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y,
                                                                 int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C4, int up) {
  const unsigned per = (unsigned)Do * Ho * Wo * C4;
  const int n = blockIdx.y;
  const f32x4* xs = x + (long)n * Di * Hi * Wi * C4;
  f32x4* ys = y + (long)n * per;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < per; idx += gridDim.x * 256u) {
    const unsigned q = idx % (unsigned)C4;
    unsigned v = idx / (unsigned)C4;
    const unsigned ox = v % (unsigned)Wo; v /= (unsigned)Wo;
    const unsigned oy = v % (unsigned)Ho;
    const unsigned oz = v / (unsigned)Ho;
    int x0, x1, y0, y1, z0 = 0, z1 = 0;
    float wx, wy, wz = 0.f;
    axis_taps((int)ox, Wi, 1, up, x0, x1, wx);
    axis_taps((int)oy, Hi, 1, up, y0, y1, wy);
    if (DIMS == 3) axis_taps((int)oz, Di, 1, up, z0, z1, wz);
    const unsigned sH = (unsigned)Wi * C4, sD = (unsigned)Hi * sH;
    const f32x4* b = xs + q;
    const f32x4 v00 = b[z0 * sD + y0 * sH + x0 * C4] * (1.f - wx) + b[z0 * sD + y0 * sH + x1 * C4] * wx;
    const f32x4 v01 = b[z0 * sD + y1 * sH + x0 * C4] * (1.f - wx) + b[z0 * sD + y1 * sH + x1 * C4] * wx;
    f32x4 r = v00 * (1.f - wy) + v01 * wy;
    if (DIMS == 3) {
      const f32x4 v10 = b[z1 * sD + y0 * sH + x0 * C4] * (1.f - wx) + b[z1 * sD + y0 * sH + x1 * C4] * wx;
      const f32x4 v11 = b[z1 * sD + y1 * sH + x0 * C4] * (1.f - wx) + b[z1 * sD + y1 * sH + x1 * C4] * wx;
      r = r * (1.f - wz) + (v10 * (1.f - wy) + v11 * wy) * wz;
    }
    ys[idx] = r;
  }
}

// adjoint of nearest x2: every input position sums the 2^dims outputs that copied it (fixed order)
__global__ void __launch_bounds__(256) resize_nearest_up_bwd_vec4_kernel(const f32x4* __restrict__ gy, f32x4* __restrict__ gx,
                                                                         int Di, int Hi, int Wi, int C4, int dims) {
  const unsigned per = (unsigned)Di * Hi * Wi * C4;
  const int n = blockIdx.y;
  const int Ho = Hi * 2, Wo = Wi * 2, Do = dims == 3 ? Di * 2 : 1;
  const f32x4* gs = gy + (long)n * Do * Ho * Wo * C4;
  f32x4* gd = gx + (long)n * per;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < per; idx += gridDim.x * 256u) {
    const unsigned q = idx % (unsigned)C4;
    unsigned v = idx / (unsigned)C4;
    const unsigned kx = v % (unsigned)Wi; v /= (unsigned)Wi;
    const unsigned ky = v % (unsigned)Hi;
    const unsigned kz = v / (unsigned)Hi;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (unsigned a = 0; a < (dims == 3 ? 2u : 1u); ++a)
      for (unsigned b = 0; b < 2u; ++b)
        for (unsigned e = 0; e < 2u; ++e) {
          const unsigned dz = dims == 3 ? 2u * kz + a : 0u;
          s += gs[((dz * (unsigned)Ho + 2u * ky + b) * (unsigned)Wo + 2u * kx + e) * (unsigned)C4 + q];
        }
    gd[idx] = s;
  }
}

// weight with which output index d reads input index k along one axis
__device__ __forceinline__ float axis_weight(int d, int k, int n_in, int n_out, int linear, int up) {
  if (d < 0 || d >= n_out) return 0.f;
  int i0, i1;
  float w1;
  axis_taps(d, n_in, linear, up, i0, i1, w1);
  float w = 0.f;
  if (i0 == k) w += 1.f - w1;
  if (i1 == k) w += w1;
  return w;
}

__global__ void __launch_bounds__(256) resize_bwd_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                         int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C,
                                                         int linear, int up, int dims) {
  const long per = (long)Di * Hi * Wi * C;
  const int n = blockIdx.y;
  const float* gs = gy + (long)n * Do * Ho * Wo * C;
  float* gd = gx + (long)n * per;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < per; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long v = idx / C;
    const int kx = (int)(v % Wi); v /= Wi;
    const int ky = (int)(v % Hi);
    const int kz = (int)(v / Hi);
    // candidate output indices per axis: up: 2k-1 .. 2k+2; down: k/2-1 .. k/2+1
    const int bx = up ? 2 * kx - 1 : kx / 2 - 1, by = up ? 2 * ky - 1 : ky / 2 - 1, bz = up ? 2 * kz - 1 : kz / 2 - 1;
    const int nc = up ? 4 : 3;
    float s = 0.f;
    for (int a = 0; a < (dims == 3 ? nc : 1); ++a) {
      const int dz = dims == 3 ? bz + a : 0;
      const float wz = dims == 3 ? axis_weight(dz, kz, Di, Do, linear, up) : 1.f;
      if (wz == 0.f) continue;
      for (int b = 0; b < nc; ++b) {
        const int dy = by + b;
        const float wy = axis_weight(dy, ky, Hi, Ho, linear, up);
        if (wy == 0.f) continue;
        for (int e = 0; e < nc; ++e) {
          const int dx = bx + e;
          const float wx = axis_weight(dx, kx, Wi, Wo, linear, up);
          if (wx == 0.f) continue;
          s += (wz * wy * wx) * gs[(((long)dz * Ho + dy) * Wo + dx) * C + c];
        }
      }
    }
    gd[idx] = s;
  }
}

// the adjoint gather with C % 4 == 0: one thread per 16 bytes of the input gradient, per-axis weights computed once
// (4 / 3 candidates per axis), float4 taps, 32-bit index math
template <int DIMS>
__global__ void __launch_bounds__(256) resize_bwd_vec4_kernel(const f32x4* __restrict__ gy, f32x4* __restrict__ gx,
                                                              int Di, int Hi, int Wi, int Do, int Ho, int Wo, int C4, int linear, int up) {
  const unsigned per = (unsigned)Di * Hi * Wi * C4;
  const int n = blockIdx.y;
  const f32x4* gs = gy + (long)n * Do * Ho * Wo * C4;
  f32x4* gd = gx + (long)n * per;
  const int nc = up ? 4 : 3;
  for (unsigned idx = blockIdx.x * 256u + threadIdx.x; idx < per; idx += gridDim.x * 256u) {
    const unsigned q = idx % (unsigned)C4;
    unsigned v = idx / (unsigned)C4;
    const int kx = (int)(v % (unsigned)Wi); v /= (unsigned)Wi;
    const int ky = (int)(v % (unsigned)Hi);
    const int kz = (int)(v / (unsigned)Hi);
    const int bx = up ? 2 * kx - 1 : kx / 2 - 1, by = up ? 2 * ky - 1 : ky / 2 - 1, bz = up ? 2 * kz - 1 : kz / 2 - 1;
    float wx[4], wy[4], wz[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wx[e] = e < nc ? axis_weight(bx + e, kx, Wi, Wo, linear, up) : 0.f;
      wy[e] = e < nc ? axis_weight(by + e, ky, Hi, Ho, linear, up) : 0.f;
      wz[e] = DIMS == 3 ? (e < nc ? axis_weight(bz + e, kz, Di, Do, linear, up) : 0.f) : (e == 0 ? 1.f : 0.f);
    }
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < (DIMS == 3 ? 4 : 1); ++a) {
      if (wz[a] == 0.f) continue;
      const int dz = DIMS == 3 ? bz + a : 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (wy[b] == 0.f) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (wx[e] == 0.f) continue;
          s += gs[(((unsigned)dz * (unsigned)Ho + (unsigned)(by + b)) * (unsigned)Wo + (unsigned)(bx + e)) * (unsigned)C4 + q] * (wz[a] * wy[b] * wx[e]);
        }
      }
    }
    gd[idx] = s;
  }
}

}  // namespace

static int resize_dims(int dims, int up, int Di, int Hi, int Wi, int& Do, int& Ho, int& Wo) {
  if (dims != 2 && dims != 3) return LF_EINVAL;
  if (dims == 2 && Di != 1) return LF_EINVAL;
  Do = dims == 3 ? (up ? Di * 2 : Di / 2) : 1;
  Ho = up ? Hi * 2 : Hi / 2;
  Wo = up ? Wi * 2 : Wi / 2;
  return (Do > 0 && Ho > 0 && Wo > 0) ? 0 : LF_EINVAL;
}

extern "C" int lf_resize_fwd(const float* x, float* y, int dims, int N, int D, int H, int W, int C, int linear, int up,
                             void* stream) {
  lf_clear_error();
  int Do, Ho, Wo;
  if (N <= 0 || C <= 0 || resize_dims(dims, up, D, H, W, Do, Ho, Wo)) return LF_EINVAL;
  const long items = (long)Do * Ho * Wo * C;
  if (!linear && (C & 3) == 0 && lf_aligned16(x) && lf_aligned16(y) && items / 4 < 0xffffffffL &&
      (long)D * H * W * C / 4 < 0xffffffffL) {
    dim3 g4((unsigned)min((items / 4 + 255) / 256, (long)65535 * 8), N);
    hipLaunchKernelGGL(resize_nearest_vec4_kernel, g4, dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)y, D, H, W, Do, Ho,
                       Wo, C / 4, up);
    return lf_launch_status();
  }
  if (linear && (C & 3) == 0 && lf_aligned16(x) && lf_aligned16(y) && items / 4 < 0xffffffffL && (long)D * H * W * C / 4 < 0xffffffffL) {
    dim3 g4((unsigned)min((items / 4 + 255) / 256, (long)65535 * 8), N);
    if (dims == 3)
      hipLaunchKernelGGL(resize_linear_vec4_kernel<3>, g4, dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)y, D, H, W, Do, Ho,
                         Wo, C / 4, up);
    else
      hipLaunchKernelGGL(resize_linear_vec4_kernel<2>, g4, dim3(256), 0, (hipStream_t)stream, (const f32x4*)x, (f32x4*)y, D, H, W, Do, Ho,
                         Wo, C / 4, up);
    return lf_launch_status();
  }
  dim3 grid((unsigned)min((items + 255) / 256, (long)65535 * 8), N);
  hipLaunchKernelGGL(resize_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, D, H, W, Do, Ho, Wo, C, 0, linear, up, dims);
  return lf_launch_status();
}

extern "C" int lf_resize_bwd(const float* gy, float* gx, int dims, int N, int D, int H, int W, int C, int linear, int up,
                             void* stream) {
  lf_clear_error();
  int Do, Ho, Wo;
  if (N <= 0 || C <= 0 || resize_dims(dims, up, D, H, W, Do, Ho, Wo)) return LF_EINVAL;
  const long items = (long)D * H * W * C;
  if (!linear && up && (C & 3) == 0 && lf_aligned16(gy) && lf_aligned16(gx) && (long)Do * Ho * Wo * C / 4 < 0xffffffffL) {
    dim3 g4((unsigned)min((items / 4 + 255) / 256, (long)65535 * 8), N);
    hipLaunchKernelGGL(resize_nearest_up_bwd_vec4_kernel, g4, dim3(256), 0, (hipStream_t)stream, (const f32x4*)gy, (f32x4*)gx, D, H, W,
                       C / 4, dims);
    return lf_launch_status();
  }
  if ((C & 3) == 0 && lf_aligned16(gy) && lf_aligned16(gx) && items / 4 < 0xffffffffL && (long)Do * Ho * Wo * C / 4 < 0xffffffffL) {
    dim3 g4((unsigned)min((items / 4 + 255) / 256, (long)65535 * 8), N);
    if (dims == 3)
      hipLaunchKernelGGL(resize_bwd_vec4_kernel<3>, g4, dim3(256), 0, (hipStream_t)stream, (const f32x4*)gy, (f32x4*)gx, D, H, W, Do, Ho, Wo,
                         C / 4, linear, up);
    else
      hipLaunchKernelGGL(resize_bwd_vec4_kernel<2>, g4, dim3(256), 0, (hipStream_t)stream, (const f32x4*)gy, (f32x4*)gx, D, H, W, Do, Ho, Wo,
                         C / 4, linear, up);
    return lf_launch_status();
  }
  dim3 grid((unsigned)min((items + 255) / 256, (long)65535 * 8), N);
  hipLaunchKernelGGL(resize_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gy, gx, D, H, W, Do, Ho, Wo, C, linear, up, dims);
  return lf_launch_status();
}
