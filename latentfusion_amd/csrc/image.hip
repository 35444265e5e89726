// Fused pose-loss kernels and the camera -> coefficient map of the render loop (gfx950).
//
//   lf_camera_coefs        Camera algebra of latentfusion/modules/geometry.py:46-590 reduced to the
//                          coefficient blocks the resampler / loss consume, evaluated in fp64 with
//                          forward-mode dual numbers so the 10-parameter Jacobian comes for free
//                          (replaces ~250 tiny ATen launches of the autograd graph per iteration).
//   lf_pose_loss_fwd/bwd   default_pose_loss (latentfusion/pose/estimation.py:70-118) fused with
//                          Photographer.interpret_logits (recon/models.py:455-484), Camera.uncrop
//                          (modules/geometry.py:261-285) and denormalize_depth (:555-558): one pass
//                          over the 480x640 frame per sample, deterministic fixed-order reductions
//                          (no float atomics: the values feed the ranking / argmin).
#include "lf_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// forward-mode duals over the 10 camera parameters (log_q 3, t 3, viewport 4)
// ---------------------------------------------------------------------------------------------
constexpr int NP = 10;
// forward-mode dual number carrying ND derivative components: ND = NP (one thread per camera, all ten derivatives) or
// ND = 1 (one thread per (camera, parameter): the same operations per component, ten times the parallelism -- the kernel is
// a serial fp64 dependency chain, 22 us for N = 8 in the first form)
template <int ND>
struct DualT {
  double v;
  double d[ND];
};
template <int ND> __device__ inline DualT<ND> dconst_(double c) { DualT<ND> r; r.v = c; for (int i = 0; i < ND; ++i) r.d[i] = 0.0; return r; }
template <int ND> __device__ inline DualT<ND> dvar_(double c, int idx, int mine) {
  DualT<ND> r = dconst_<ND>(c);
  if (ND == 1) r.d[0] = (idx == mine) ? 1.0 : 0.0; else r.d[idx % ND] = 1.0;
  return r;
}
template <int ND> __device__ inline DualT<ND> operator+(const DualT<ND>& a, const DualT<ND>& b) { DualT<ND> r; r.v = a.v + b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
template <int ND> __device__ inline DualT<ND> operator-(const DualT<ND>& a, const DualT<ND>& b) { DualT<ND> r; r.v = a.v - b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
template <int ND> __device__ inline DualT<ND> operator-(const DualT<ND>& a) { DualT<ND> r; r.v = -a.v; for (int i = 0; i < ND; ++i) r.d[i] = -a.d[i]; return r; }
template <int ND> __device__ inline DualT<ND> operator*(const DualT<ND>& a, const DualT<ND>& b) { DualT<ND> r; r.v = a.v * b.v; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
template <int ND> __device__ inline DualT<ND> operator*(const DualT<ND>& a, double s) { DualT<ND> r; r.v = a.v * s; for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * s; return r; }
template <int ND> __device__ inline DualT<ND> operator+(const DualT<ND>& a, double s) { DualT<ND> r = a; r.v += s; return r; }
template <int ND> __device__ inline DualT<ND> operator-(const DualT<ND>& a, double s) { DualT<ND> r = a; r.v -= s; return r; }
template <int ND> __device__ inline DualT<ND> operator/(const DualT<ND>& a, const DualT<ND>& b) {
  DualT<ND> r; const double inv = 1.0 / b.v; r.v = a.v * inv;
  for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv;
  return r;
}
template <int ND> __device__ inline DualT<ND> dsqrt(const DualT<ND>& a) {
  DualT<ND> r; r.v = sqrt(a.v); const double k = r.v > 0.0 ? 0.5 / r.v : 0.0;      // subgradient 0 at 0 (torch.norm)
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * k;
  return r;
}
template <int ND> __device__ inline DualT<ND> dsin(const DualT<ND>& a) { DualT<ND> r; r.v = sin(a.v); const double c = cos(a.v); for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * c; return r; }
template <int ND> __device__ inline DualT<ND> dcos(const DualT<ND>& a) { DualT<ND> r; r.v = cos(a.v); const double s = -sin(a.v); for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * s; return r; }
// max(a, floor): derivative passes only where a is the active branch (torch.clamp(min=))
template <int ND> __device__ inline DualT<ND> dclamp_min(const DualT<ND>& a, double lo) { return a.v >= lo ? a : dconst_<ND>(lo); }

constexpr int NOUT = 24;   // 18 O2C coefficients + (ax, bx, ay, by) of the crop->frame map + (a_depth, b_depth)

template <int ND>
__global__ void camera_coefs_kernel(const float* __restrict__ params, const float* __restrict__ intr,
                                    float cube, float z_span, int crop_h, int crop_w,
                                    float* __restrict__ out, float* __restrict__ jac, int N) {
  typedef DualT<ND> Dual;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = ND == 1 ? tid / NP : tid, mine = ND == 1 ? tid % NP : 0;
  if (n >= N) return;
  auto dconst = [](double c) { return dconst_<ND>(c); };
  auto dvar = [&](double c, int idx) { return dvar_<ND>(c, idx, mine); };
  const float* p = params + n * NP;
  Dual w[3], t[3], vp[4];
  for (int i = 0; i < 3; ++i) w[i] = dvar((double)p[i], i);
  for (int i = 0; i < 3; ++i) t[i] = dvar((double)p[3 + i], 3 + i);
  for (int i = 0; i < 4; ++i) vp[i] = dvar((double)p[6 + i], 6 + i);
  const double fu = intr[n * 4 + 0], fv = intr[n * 4 + 1], u0 = intr[n * 4 + 2], v0 = intr[n * 4 + 3];

  // quaternion = qexp(log_q) (three/quaternion.py:287-311); R = quat_to_mat(normalize(normalize(q)))
  const Dual theta = dsqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  const Dual sc = dsin(theta) / dclamp_min(theta, 1e-8);
  Dual q[4] = {dcos(theta), sc * w[0], sc * w[1], sc * w[2]};
  for (int rep = 0; rep < 2; ++rep) {
    const Dual nrm = dclamp_min(dsqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), 1e-12);
    for (int i = 0; i < 4; ++i) q[i] = q[i] / nrm;
  }
  const Dual qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  const Dual tx = qx * 2.0, ty = qy * 2.0, tz = qz * 2.0;
  Dual R[3][3];
  R[0][0] = dconst(1.0) - (ty * qy + tz * qz); R[0][1] = ty * qx - tz * qw;               R[0][2] = tz * qx + ty * qw;
  R[1][0] = ty * qx + tz * qw;               R[1][1] = dconst(1.0) - (tx * qx + tz * qz); R[1][2] = tz * qy - tx * qw;
  R[2][0] = tz * qx - ty * qw;               R[2][1] = tz * qy + tx * qw;               R[2][2] = dconst(1.0) - (tx * qx + ty * qy);

  // O2C: grid = s * R^T (p_cam - t), p_cam = ((u-u0)/fu z, (v-v0)/fv z, z)   (geometry.py:469-531,669-686)
  const Dual vw = vp[2] - vp[0], vh = vp[3] - vp[1];
  const Dual al0 = (vp[0] - u0) * (1.0 / fu), al1 = vw * (1.0 / fu);
  const Dual be0 = (vp[1] - v0) * (1.0 / fv), be1 = vh * (1.0 / fv);
  const Dual g0 = t[2] - (double)z_span;
  const double g1 = z_span, s = 2.0 / cube;
  Dual c[6][3];
  for (int r = 0; r < 3; ++r) {
    const Dual m0 = R[0][r], m1 = R[1][r], m2 = R[2][r];        // column j of R^T = row j of R
    const Dual rt = m0 * t[0] + m1 * t[1] + m2 * t[2];           // (R^T t)_r
    c[0][r] = (al0 * g0 * m0 + be0 * g0 * m1 + g0 * m2 - rt) * s;
    c[1][r] = (al1 * g0 * m0) * s;
    c[2][r] = (be1 * g0 * m1) * s;
    c[3][r] = (al0 * m0 + be0 * m1 + m2) * (g1 * s);
    c[4][r] = (al1 * m0) * (g1 * s);
    c[5][r] = (be1 * m1) * (g1 * s);
  }
  Dual o[NOUT];
  for (int j = 0; j < 6; ++j) for (int r = 0; r < 3; ++r) o[j * 3 + r] = c[j][r];
  // crop -> frame sampling map: ix = ax * x + bx  (uncrop, geometry.py:261-285; align_corners=False)
  o[18] = dconst((double)crop_w) / vw;
  o[19] = -(vp[0] * o[18]) - 0.5;
  o[20] = dconst((double)crop_h) / vh;
  o[21] = -(vp[1] * o[20]) - 0.5;
  // denormalize_depth: z = d * a + b, a = (zfar - znear + 0.02) / 2, b = (zfar + znear) / 2 = t_z
  o[22] = dconst((double)z_span + 0.01);
  o[23] = t[2];
  for (int j = 0; j < NOUT; ++j) {
    if (ND != 1 || mine == 0) out[n * NOUT + j] = (float)o[j].v;
    if (ND == 1) jac[(n * NOUT + j) * NP + mine] = (float)o[j].d[0];
    else for (int i = 0; i < ND; ++i) jac[(n * NOUT + j) * NP + i] = (float)o[j].d[i];
  }
}

// gparams[n][i] = sum_j gout[n][j] * jac[n][j][i]   (fixed order)
__global__ void camera_coefs_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ jac,
                                        float* __restrict__ gparams, int N) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= N * NP) return;
  const int n = idx / NP, i = idx % NP;
  double s = 0.0;
  for (int j = 0; j < NOUT; ++j) s += (double)gout[n * NOUT + j] * (double)jac[(n * NOUT + j) * NP + i];
  gparams[idx] = (float)s;
}

// ---------------------------------------------------------------------------------------------
// pose loss
// ---------------------------------------------------------------------------------------------
constexpr int NSUM = 8;     // S0 sum l1, S1 sum l1*sig*mt, S2 sum sig*mt, S3 sum sig, S4 sum sig*mt*valid,
                            // S5 sum mt*valid, S6 sum bce, S7 unused
constexpr int LOSS_BLOCK = 256;

struct PixelFwd {
  float sig, dhat, pd, td, mt, valid, l1, logit;
  float wx0, wx1, wy0, wy1, mx, my;     // bilinear weights and border-clip masks of the logit sample
  int x0, x1, y0, y1, xn, yn;           // bilinear corners, nearest corner
  float dn;                             // normalised depth of the nearest crop pixel (after apply_mask)
  bool mask_on;                         // mask(nearest) > 0.5 (apply_mask gate)
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ void clip_pos(float p, int size, float& pos, float& mult) {
  const float hi = (float)(size - 1);
  mult = (p > 0.f && p < hi) ? 1.f : 0.f;
  pos = fminf(fmaxf(p, 0.f), hi);
}

// logits: [N][h*w][2] = (depth_logit, mask_logit) per crop pixel (channels-last head output)
// MASKED: the form the cross-entropy / Metropolis estimators feed the loss (reference pose/estimation.py:207-216): the
// de-normalised crop depth is multiplied by the crop's own sigmoid mask BEFORE it is uncropped (the gradient estimator
// passes the depth without that factor, :703-713)
// MODE 0: head logits as above.  MODE 1 = MASKED.  MODE 2 (round 5; the MODULE path's default_pose_loss, which is handed the
// metric depth crop itself -- already de-normalised / gated by the caller, reference pose/estimation.py:703-713): channel 0
// of `lg` IS the predicted depth; tanh, apply_mask and denormalize_depth are skipped and have no gradient here.
template <int MODE = 0>
__device__ __forceinline__ PixelFwd pixel_forward(const float* __restrict__ lg, const float* __restrict__ cf,
                                                  int h, int w, int x, int y, float td_raw, float mt) {
  PixelFwd r;
  float px, py;
  clip_pos(cf[0] * (float)x + cf[1], w, px, r.mx);
  clip_pos(cf[2] * (float)y + cf[3], h, py, r.my);
  const float fx = floorf(px), fy = floorf(py);
  r.wx1 = px - fx; r.wx0 = 1.f - r.wx1; r.wy1 = py - fy; r.wy0 = 1.f - r.wy1;
  r.x0 = (int)fx; r.y0 = (int)fy; r.x1 = min(r.x0 + 1, w - 1); r.y1 = min(r.y0 + 1, h - 1);
  r.xn = (int)nearbyintf(px); r.yn = (int)nearbyintf(py);          // ATen nearest: round half to even
  const float m00 = lg[(r.y0 * w + r.x0) * 2 + 1], m01 = lg[(r.y0 * w + r.x1) * 2 + 1];
  const float m10 = lg[(r.y1 * w + r.x0) * 2 + 1], m11 = lg[(r.y1 * w + r.x1) * 2 + 1];
  r.logit = m00 * (r.wx0 * r.wy0) + m01 * (r.wx1 * r.wy0) + m10 * (r.wx0 * r.wy1) + m11 * (r.wx1 * r.wy1);
  r.sig = sigmoidf_(r.logit);
  const float dl = lg[(r.yn * w + r.xn) * 2 + 0], ml = lg[(r.yn * w + r.xn) * 2 + 1];
  if constexpr (MODE == 2) {
    r.mask_on = true;
    r.dn = dl;
    r.dhat = dl;
  } else {
    r.mask_on = sigmoidf_(ml) > 0.5f;
    r.dn = r.mask_on ? tanhf(dl) : -1.f;                             // (tanh+1)*(mask>0.5)-1
    r.dhat = r.dn * cf[4] + cf[5];
    if constexpr (MODE == 1) r.dhat *= sigmoidf_(ml);
  }
  r.pd = r.dhat * r.sig;
  r.mt = mt;
  r.valid = (td_raw == 0.f && mt > 0.1f) ? 0.f : 1.f;
  r.td = td_raw * mt;
  r.l1 = fabsf(r.pd - r.td) * r.valid;
  return r;
}

__device__ __forceinline__ void block_reduce_store(float (&acc)[NSUM], float* __restrict__ dst) {
  __shared__ float red[LOSS_BLOCK / 64][NSUM];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NSUM; ++i) {
    const float s = lf_wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < NSUM) {
    float s = 0.f;
    for (int wv = 0; wv < LOSS_BLOCK / 64; ++wv) s += red[wv][threadIdx.x];
    dst[threadIdx.x] = s;
  }
}

template <int MODE>
__global__ void __launch_bounds__(LOSS_BLOCK) pose_loss_fwd_kernel(
    const float* __restrict__ logits, const float* __restrict__ coef, const float* __restrict__ tdepth,
    const float* __restrict__ tmask, float* __restrict__ partial, int nblk, int h, int w, int H, int W) {
  const int n = blockIdx.y;
  const float* lg = logits + (long)n * h * w * 2;
  const float* cf = coef + n * NOUT + 18;
  float acc[NSUM];
#pragma unroll
  for (int i = 0; i < NSUM; ++i) acc[i] = 0.f;
  for (int p = blockIdx.x * LOSS_BLOCK + threadIdx.x; p < H * W; p += nblk * LOSS_BLOCK) {
    const int y = p / W, x = p - y * W;
    const PixelFwd f = pixel_forward<MODE>(lg, cf, h, w, x, y, tdepth[p], tmask[p]);
    acc[0] += f.l1;
    acc[1] += f.l1 * (f.sig * f.mt);
    acc[2] += f.sig * f.mt;
    acc[3] += f.sig;
    acc[4] += f.sig * (f.mt * f.valid);
    acc[5] += f.mt * f.valid;
    // BCE with logits, numerically stable form used by ATen: (1 - t) * x + log1p(exp(-|x|)) + max(-x, 0)
    acc[6] += (1.f - f.mt) * f.logit + (log1pf(expf(-fabsf(f.logit))) + fmaxf(-f.logit, 0.f));
  }
  block_reduce_store(acc, partial + ((long)n * nblk + blockIdx.x) * NSUM);
}

// second stage: sums -> the four losses, their weighted total and d(mean total)/d(sums)
__global__ void pose_loss_finish_kernel(const float* __restrict__ partial, int nblk, const float* __restrict__ weights,
                                        int N, int HW, float* __restrict__ sums, float* __restrict__ losses,
                                        float* __restrict__ gsums) {
  const int n = blockIdx.x;
  __shared__ double S[NSUM];
  // one wave: lane l adds the partials of blocks l, l + 64, ... in fp64, then a fixed butterfly over the lanes (a serial walk
  // over the 240 blocks by one lane per component was 240 dependent global loads = 28 us of pure latency)
#pragma unroll
  for (int c = 0; c < NSUM; ++c) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)partial[((long)n * nblk + b) * NSUM + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) {
      S[c] = s;
      sums[n * NSUM + c] = (float)s;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float S0 = (float)S[0], S1 = (float)S[1], S2 = (float)S[2], S3 = (float)S[3], S4 = (float)S[4],
                S5 = (float)S[5], S6 = (float)S[6];
    const float w_depth = weights[0], w_ov = weights[1], w_iou = weights[2], w_mask = weights[3];
    const float inv_hw = 1.f / (float)HW;
    const float depth = S0 * inv_hw;
    const float num = fmaxf(S1, 1e-5f), den = fmaxf(S2, 1e-4f);
    const float ov = num / den;
    const float uni = S3 + S5 - S4;
    const float iou = logf(fmaxf(uni, 1e-4f)) - logf(fmaxf(S4, 1e-4f));
    const float mask = S6 * inv_hw;
    losses[n * 8 + 0] = depth; losses[n * 8 + 1] = ov; losses[n * 8 + 2] = iou; losses[n * 8 + 3] = mask;
    losses[n * 8 + 4] = w_depth * depth + w_ov * ov + w_iou * iou + w_mask * mask;      // rank / optim loss
    losses[n * 8 + 5] = 0.f; losses[n * 8 + 6] = 0.f; losses[n * 8 + 7] = 0.f;
    const float k = 1.f / (float)N;                                  // optimised quantity = mean over samples
    float* g = gsums + n * NSUM;
    const float d_uni = uni > 1e-4f ? 1.f / uni : 0.f;
    g[0] = k * w_depth * inv_hw;
    g[1] = k * w_ov * (S1 > 1e-5f ? 1.f / den : 0.f);
    g[2] = k * w_ov * (S2 > 1e-4f ? -num / (den * den) : 0.f);
    g[3] = k * w_iou * d_uni;
    g[4] = k * w_iou * (-d_uni - (S4 > 1e-4f ? 1.f / S4 : 0.f));
    g[5] = 0.f;
    g[6] = k * w_mask * inv_hw;
    g[7] = 0.f;
  }
}

// backward stage A: per frame pixel, d/d(sampled depth) and d/d(sampled logit) + coefficient grads
template <int MODE>
__global__ void __launch_bounds__(LOSS_BLOCK) pose_loss_bwd_pixels_kernel(
    const float* __restrict__ logits, const float* __restrict__ coef, const float* __restrict__ tdepth,
    const float* __restrict__ tmask, const float* __restrict__ gsums, float* __restrict__ gd_frame,
    float* __restrict__ gm_frame, float* __restrict__ partial, int nblk, int h, int w, int H, int W) {
  const int n = blockIdx.y;
  const float* lg = logits + (long)n * h * w * 2;
  const float* cf = coef + n * NOUT + 18;
  const float* g = gsums + n * NSUM;
  const float g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3], g4 = g[4], g6 = g[6];
  float acc[NSUM];     // 0: d/d ax, 1: d/d bx, 2: d/d ay, 3: d/d by, 4: d/d a_depth, 5: d/d b_depth
#pragma unroll
  for (int i = 0; i < NSUM; ++i) acc[i] = 0.f;
  for (int p = blockIdx.x * LOSS_BLOCK + threadIdx.x; p < H * W; p += nblk * LOSS_BLOCK) {
    const int y = p / W, x = p - y * W;
    const PixelFwd f = pixel_forward<MODE>(lg, cf, h, w, x, y, tdepth[p], tmask[p]);
    const float dl1 = g0 + g1 * (f.sig * f.mt);
    const float diff = f.pd - f.td;
    const float dpd = dl1 * f.valid * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
    const float ddhat = dpd * f.sig;
    const float dsig = dpd * f.dhat + g1 * f.l1 * f.mt + g2 * f.mt + g3 + g4 * (f.mt * f.valid);
    const float dlogit = dsig * f.sig * (1.f - f.sig) + g6 * (f.sig - f.mt);
    gd_frame[(long)n * H * W + p] = ddhat;
    gm_frame[(long)n * H * W + p] = dlogit;
    // d logit / d ix, d logit / d iy of the bilinear sample (border clip masks the gradient)
    const float m00 = lg[(f.y0 * w + f.x0) * 2 + 1], m01 = lg[(f.y0 * w + f.x1) * 2 + 1];
    const float m10 = lg[(f.y1 * w + f.x0) * 2 + 1], m11 = lg[(f.y1 * w + f.x1) * 2 + 1];
    const float dix = ((m01 - m00) * f.wy0 + (m11 - m10) * f.wy1) * f.mx * dlogit;
    const float diy = ((m10 - m00) * f.wx0 + (m11 - m01) * f.wx1) * f.my * dlogit;
    acc[0] += dix * (float)x; acc[1] += dix;
    acc[2] += diy * (float)y; acc[3] += diy;
    if constexpr (MODE != 2) { acc[4] += ddhat * f.dn;   acc[5] += ddhat; }
  }
  block_reduce_store(acc, partial + ((long)n * nblk + blockIdx.x) * NSUM);
}

// backward stage B (rows): adjoint of the separable sampling along x, for every frame row y
//   Td[n][y][cx] = sum_x [nearest(x) == cx] gd[y][x]      Tm[n][y][cx] = sum_x wx(x, cx) gm[y][x]
__global__ void __launch_bounds__(256) pose_loss_bwd_rows_kernel(
    const float* __restrict__ coef, const float* __restrict__ gd_frame, const float* __restrict__ gm_frame,
    float* __restrict__ Td, float* __restrict__ Tm, int w, int H, int W) {
  const int n = blockIdx.z;
  const int y = blockIdx.y;
  const int cx = blockIdx.x * blockDim.x + threadIdx.x;
  if (cx >= w) return;
  const float ax = coef[n * NOUT + 18], bx = coef[n * NOUT + 19];
  // frame columns whose clipped sample position can touch crop column cx: ix in (cx-1, cx+1); the
  // border columns additionally collect everything clipped onto them
  int xlo = (int)floorf(((float)cx - 1.f - bx) / ax) - 1, xhi = (int)ceilf(((float)cx + 1.f - bx) / ax) + 1;
  if (cx == 0) xlo = 0;
  if (cx == w - 1) xhi = W - 1;
  xlo = max(xlo, 0); xhi = min(xhi, W - 1);
  const float* gd = gd_frame + ((long)n * H + y) * W;
  const float* gm = gm_frame + ((long)n * H + y) * W;
  float sd = 0.f, sm = 0.f;
  for (int x = xlo; x <= xhi; ++x) {
    float px, mult;
    clip_pos(ax * (float)x + bx, w, px, mult);
    const float fx = floorf(px);
    const int x0 = (int)fx, x1 = min(x0 + 1, w - 1);
    const float wx1 = px - fx, wx0 = 1.f - wx1;
    float wgt = 0.f;
    if (x0 == cx) wgt += wx0;
    if (x1 == cx) wgt += wx1;
    sm += wgt * gm[x];
    if ((int)nearbyintf(px) == cx) sd += gd[x];
  }
  Td[((long)n * H + y) * w + cx] = sd;
  Tm[((long)n * H + y) * w + cx] = sm;
}

// backward stage C (columns): finish along y, apply the tanh / sigmoid / apply_mask chain, write
// d(loss)/d(head logits) [N][h*w][2]  (MODE 2: channel 0 is d(loss)/d(depth crop) itself)
template <int MODE>
__global__ void __launch_bounds__(256) pose_loss_bwd_cols_kernel(
    const float* __restrict__ logits, const float* __restrict__ coef, const float* __restrict__ Td,
    const float* __restrict__ Tm, float* __restrict__ glogits, int h, int w, int H) {
  const int n = blockIdx.z;
  const int cy = blockIdx.y;
  const int cx = blockIdx.x * blockDim.x + threadIdx.x;
  if (cx >= w) return;
  const float ay = coef[n * NOUT + 20], by = coef[n * NOUT + 21], a_depth = coef[n * NOUT + 22];
  int ylo = (int)floorf(((float)cy - 1.f - by) / ay) - 1, yhi = (int)ceilf(((float)cy + 1.f - by) / ay) + 1;
  if (cy == 0) ylo = 0;
  if (cy == h - 1) yhi = H - 1;
  ylo = max(ylo, 0); yhi = min(yhi, H - 1);
  float sd = 0.f, sm = 0.f;
  for (int y = ylo; y <= yhi; ++y) {
    float py, mult;
    clip_pos(ay * (float)y + by, h, py, mult);
    const float fy = floorf(py);
    const int y0 = (int)fy, y1 = min(y0 + 1, h - 1);
    const float wy1 = py - fy, wy0 = 1.f - wy1;
    float wgt = 0.f;
    if (y0 == cy) wgt += wy0;
    if (y1 == cy) wgt += wy1;
    sm += wgt * Tm[((long)n * H + y) * w + cx];
    if ((int)nearbyintf(py) == cy) sd += Td[((long)n * H + y) * w + cx];
  }
  const long o = ((long)n * h * w + (long)cy * w + cx) * 2;
  if constexpr (MODE == 2) {
    glogits[o] = sd;
  } else {
    const float dl = logits[o], ml = logits[o + 1];
    const float th = tanhf(dl);
    const bool on = sigmoidf_(ml) > 0.5f;
    glogits[o] = on ? sd * a_depth * (1.f - th * th) : 0.f;     // d dhat/d dn = a_depth; (mask>0.5) gate has no gradient
  }
  glogits[o + 1] = sm;
}

__global__ void reduce_partials_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ out, int ncomp,
                                       int out_stride, int out_offset) {
  const int n = blockIdx.x;
  for (int c = 0; c < ncomp; ++c) {                               // (one wave; lanes over the blocks, fixed butterfly: see above)
    double s = 0.0;
    for (int b = threadIdx.x; b < nblk; b += 64) s += (double)partial[((long)n * nblk + b) * NSUM + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (threadIdx.x == 0) out[n * out_stride + out_offset + c] = (float)s;
  }
}

constexpr int LOSS_NBLK = 240;    // blocks per sample over the 480x640 frame (5 pixels per thread)

}  // namespace

extern "C" int lf_camera_coefs(const float* params, const float* intrinsics, float cube_size, float z_span,
                               int crop_h, int crop_w, float* coefs, float* jac, int N, void* stream) {
  lf_clear_error();
  if (N <= 0 || crop_h <= 0 || crop_w <= 0 || cube_size <= 0.f) return LF_EINVAL;
  // one thread per (camera, parameter)
  hipLaunchKernelGGL((camera_coefs_kernel<1>), dim3((N * NP + 63) / 64), dim3(64), 0, (hipStream_t)stream, params, intrinsics,
                     cube_size, z_span, crop_h, crop_w, coefs, jac, N);
  return lf_launch_status();
}

extern "C" int lf_camera_coefs_bwd(const float* gcoefs, const float* jac, float* gparams, int N, void* stream) {
  lf_clear_error();
  if (N <= 0) return LF_EINVAL;
  hipLaunchKernelGGL(camera_coefs_bwd_kernel, dim3((N * NP + 63) / 64), dim3(64), 0, (hipStream_t)stream, gcoefs, jac,
                     gparams, N);
  return lf_launch_status();
}

extern "C" size_t lf_pose_loss_scratch_bytes(int N, int h, int w, int H, int W) {
  // partial sums + two frame-sized gradient maps + two row-pass intermediates
  return ((size_t)N * LOSS_NBLK * NSUM + 2 * (size_t)N * H * W + 2 * (size_t)N * H * w) * sizeof(float);
}

static int pose_loss_fwd_launch(int mode, const float* logits, const float* coefs, const float* target_depth,
                                const float* target_mask, const float* weights, float* sums, float* losses,
                                float* gsums, void* scratch, size_t scratch_bytes,
                                int N, int h, int w, int H, int W, void* stream) {
  lf_clear_error();
  if (N <= 0 || h <= 1 || w <= 1 || H <= 0 || W <= 0) return LF_EINVAL;
  if (scratch_bytes < lf_pose_loss_scratch_bytes(N, h, w, H, W)) return LF_ENOSPC;
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)scratch;
  if (mode == 1)
    hipLaunchKernelGGL(pose_loss_fwd_kernel<1>, dim3(LOSS_NBLK, N), dim3(LOSS_BLOCK), 0, s, logits, coefs, target_depth,
                       target_mask, partial, LOSS_NBLK, h, w, H, W);
  else if (mode == 2)
    hipLaunchKernelGGL(pose_loss_fwd_kernel<2>, dim3(LOSS_NBLK, N), dim3(LOSS_BLOCK), 0, s, logits, coefs, target_depth,
                       target_mask, partial, LOSS_NBLK, h, w, H, W);
  else
    hipLaunchKernelGGL(pose_loss_fwd_kernel<0>, dim3(LOSS_NBLK, N), dim3(LOSS_BLOCK), 0, s, logits, coefs, target_depth,
                       target_mask, partial, LOSS_NBLK, h, w, H, W);
  int st = lf_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(pose_loss_finish_kernel, dim3(N), dim3(64), 0, s, partial, LOSS_NBLK, weights, N, H * W, sums, losses,
                     gsums);
  return lf_launch_status();
}

extern "C" int lf_pose_loss_fwd(const float* logits, const float* coefs, const float* target_depth,
                                const float* target_mask, const float* weights, float* sums, float* losses,
                                float* gsums, void* scratch, size_t scratch_bytes,
                                int N, int h, int w, int H, int W, void* stream) {
  return pose_loss_fwd_launch(0, logits, coefs, target_depth, target_mask, weights, sums, losses, gsums, scratch,
                              scratch_bytes, N, h, w, H, W, stream);
}

extern "C" int lf_pose_loss_fwd_depth(const float* depth_and_logits, const float* coefs, const float* target_depth,
                                      const float* target_mask, const float* weights, float* sums, float* losses,
                                      float* gsums, void* scratch, size_t scratch_bytes,
                                      int N, int h, int w, int H, int W, void* stream) {
  return pose_loss_fwd_launch(2, depth_and_logits, coefs, target_depth, target_mask, weights, sums, losses, gsums, scratch,
                              scratch_bytes, N, h, w, H, W, stream);
}

extern "C" int lf_pose_loss_fwd_masked(const float* logits, const float* coefs, const float* target_depth,
                                       const float* target_mask, const float* weights, float* sums, float* losses,
                                       void* scratch, size_t scratch_bytes,
                                       int N, int h, int w, int H, int W, void* stream) {
  // forward only (the estimators that use this form do not differentiate): gsums lands in the scratch tail
  float* gs = (float*)scratch + (size_t)N * LOSS_NBLK * NSUM;
  return pose_loss_fwd_launch(1, logits, coefs, target_depth, target_mask, weights, sums, losses, gs, scratch,
                              scratch_bytes, N, h, w, H, W, stream);
}

static int pose_loss_bwd_launch(int mode, const float* logits, const float* coefs, const float* target_depth,
                                const float* target_mask, const float* gsums, float* glogits, float* gcoefs,
                                void* scratch, size_t scratch_bytes, int N, int h, int w, int H, int W, void* stream);

extern "C" int lf_pose_loss_bwd(const float* logits, const float* coefs, const float* target_depth,
                                const float* target_mask, const float* gsums, float* glogits, float* gcoefs,
                                void* scratch, size_t scratch_bytes, int N, int h, int w, int H, int W, void* stream) {
  return pose_loss_bwd_launch(0, logits, coefs, target_depth, target_mask, gsums, glogits, gcoefs, scratch, scratch_bytes, N, h, w, H, W,
                              stream);
}

extern "C" int lf_pose_loss_bwd_depth(const float* depth_and_logits, const float* coefs, const float* target_depth,
                                      const float* target_mask, const float* gsums, float* gcrop, float* gcoefs,
                                      void* scratch, size_t scratch_bytes, int N, int h, int w, int H, int W, void* stream) {
  return pose_loss_bwd_launch(2, depth_and_logits, coefs, target_depth, target_mask, gsums, gcrop, gcoefs, scratch, scratch_bytes, N, h, w,
                              H, W, stream);
}

static int pose_loss_bwd_launch(int mode, const float* logits, const float* coefs, const float* target_depth,
                                const float* target_mask, const float* gsums, float* glogits, float* gcoefs,
                                void* scratch, size_t scratch_bytes, int N, int h, int w, int H, int W, void* stream) {
  lf_clear_error();
  if (N <= 0 || h <= 1 || w <= 1 || H <= 0 || W <= 0) return LF_EINVAL;
  if (scratch_bytes < lf_pose_loss_scratch_bytes(N, h, w, H, W)) return LF_ENOSPC;
  hipStream_t s = (hipStream_t)stream;
  float* partial = (float*)scratch;
  float* gd = partial + (size_t)N * LOSS_NBLK * NSUM;
  float* gm = gd + (size_t)N * H * W;
  float* Td = gm + (size_t)N * H * W;
  float* Tm = Td + (size_t)N * H * w;
  if (mode == 2)
    hipLaunchKernelGGL(pose_loss_bwd_pixels_kernel<2>, dim3(LOSS_NBLK, N), dim3(LOSS_BLOCK), 0, s, logits, coefs, target_depth,
                       target_mask, gsums, gd, gm, partial, LOSS_NBLK, h, w, H, W);
  else
    hipLaunchKernelGGL(pose_loss_bwd_pixels_kernel<0>, dim3(LOSS_NBLK, N), dim3(LOSS_BLOCK), 0, s, logits, coefs, target_depth,
                       target_mask, gsums, gd, gm, partial, LOSS_NBLK, h, w, H, W);
  int st = lf_launch_status();
  if (st) return st;
  // gcoefs[n][18..23] <- reduced (ax, bx, ay, by, a_depth, b_depth) gradients; [0..17] are left to the resampler
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(N), dim3(64), 0, s, partial, LOSS_NBLK, gcoefs, 6, NOUT, 18);
  hipLaunchKernelGGL(pose_loss_bwd_rows_kernel, dim3((w + 255) / 256, H, N), dim3(256), 0, s, coefs, gd, gm, Td, Tm, w, H, W);
  if (mode == 2)
    hipLaunchKernelGGL(pose_loss_bwd_cols_kernel<2>, dim3((w + 255) / 256, h, N), dim3(256), 0, s, logits, coefs, Td, Tm, glogits,
                       h, w, H);
  else
    hipLaunchKernelGGL(pose_loss_bwd_cols_kernel<0>, dim3((w + 255) / 256, h, N), dim3(256), 0, s, logits, coefs, Td, Tm, glogits,
                       h, w, H);
  return lf_launch_status();
}
