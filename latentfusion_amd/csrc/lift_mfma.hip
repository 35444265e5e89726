// The 2-D -> 3-D lift of the TRAINING step (BASELINE cfg 5; reference modules/geometry.py:711-731 FactorProjection2d3d under the
// bf16 autocast policy: pointwise conv C_img -> C0*S, LeakyReLU, PixelNorm over all C0*S channels, view as (C0, S) -> a
// channels-last volume) as ONE kernel each way, for C_img = C0 = 16.
//
// Round 5 ran it as conv1x1 -> fp32 rows (V*P x 2048: 4.3 GB at 32 views of 128^2) -> lf_lift_norm_unfold forward and
// lf_lift_bwd -> fp32 rows -> data-gradient conv1x1 + generic weight gradient + column sums backward: 2.9 + 7.7 ms of the
// step, all of it moving those rows.  But the contraction is K = 16: recomputing a row costs nothing next to writing it.
//   forward   per 16 pixels and depth d: ONE v_mfma_f32_16x16x16_bf16 gives the 16 channels (c, d) of the 16 pixels = the 16
//             records the volume wants for that depth; PixelNorm needs the sum over all (c, d) first, so the row is formed
//             twice (pass 1: sum of squares, pass 2: scale and store bf16 records) -- 2 x S MFMAs instead of 8 KB of row.
//   backward  PixelNorm' / LeakyReLU' from the gradient volume and the saved volume per record, then per depth d three uses of
//             the 16 x 16 block gp[c][pixel]: gx[cin][pixel] += W_d^T gp (MFMA, K = c), gW_d[c][cin] += gp x^T (MFMA, K =
//             pixel; gp and x reach the K-major operand layout through the LDS transpose load ds_read_b64_tr_b16) and the
//             bias gradient (row sums of the same transposed operand).  A workgroup is 8 waves, wave w owns depths
//             [w S/8, (w+1) S/8) of every 16-pixel group the workgroup walks (its gW / bias accumulators stay in registers for
//             the whole launch); the per-pixel dot of PixelNorm' and gx are summed across the waves through LDS in wave order.
//             Per-workgroup partials of gW / bias are summed by a second kernel in workgroup order: deterministic.
// Operands are bf16 (x, W, gp are bf16 VALUES under the policy: the products are exact), accumulation fp32.
#include "lf_common.h"

namespace {

typedef __bf16 bf16x4m __attribute__((ext_vector_type(4)));
typedef short s16x4m __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bf16x4m lds_bf16x4m;
typedef unsigned u32x2m __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma16(const bf16x4m a, const bf16x4m b, const f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4m, a), __builtin_bit_cast(s16x4m, b), c, 0, 0, 0);
}
__device__ __forceinline__ float quad_sum(float v) {               // over the four lanes n, n+16, n+32, n+48
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}

// wtab: bf16 [S][16 c][16 cin], entry = W[c*S + d][cin]; btab: fp32 [S][16 c] (or NULL).  x: fp32 rows [R][16].
// vol: bf16 records, record (v, d, p) at ((v*S + d)*P + p); norm: [R].  R % 16 == 0, P % 16 == 0.
__global__ void __launch_bounds__(256, 2) lift_fwd_mfma_kernel(const float* __restrict__ x, const __bf16* __restrict__ wtab,
                                                              const float* __restrict__ btab, unsigned char* __restrict__ vol,
                                                              float* __restrict__ norm, long R, long P, int S, float he, float slope,
                                                              float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lw = smem;                                       // S * 512 B
  float* lb = (float*)(smem + (size_t)S * 512);                   // S * 16 floats
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kg = lane >> 4;
  for (int i = tid; i < S * 32; i += 256) ((f32x4*)lw)[i] = ((const f32x4*)wtab)[i];
  for (int i = tid; i < S * 16; i += 256) lb[i] = btab ? btab[i] : 0.f;
  __syncthreads();
  const long groups = R / 16;
  const float inv_cs = 1.f / (16.f * (float)S);
  for (long g = (long)blockIdx.x * 4 + wave; g < groups; g += (long)gridDim.x * 4) {
    const long pix = g * 16 + n;
    const long v = pix / P, p = pix - v * P;
    const bf16x4m b = __builtin_convertvector(*(const f32x4*)(x + pix * 16 + kg * 4), bf16x4m);
    const unsigned char* aw = lw + (n * 16 + kg * 4) * 2;          // A operand: row c = n, cin 4 kg .. +3 (+ d * 512)
    const float* bw = lb + kg * 4;                                 // bias of channels c = 4 kg .. +3 (+ d * 16)
    float ss = 0.f;
    for (int d = 0; d < S; ++d) {
      const f32x4 acc = mfma16(*(const bf16x4m*)(aw + d * 512), b, (f32x4){0.f, 0.f, 0.f, 0.f});
      const f32x4 bv = *(const f32x4*)(bw + d * 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = __builtin_fmaf(acc[e], he, bv[e]);
        t = fmaxf(t, t * slope);
        ss = __builtin_fmaf(t, t, ss);
      }
    }
    ss = quad_sum(ss);
    const float rn = sqrtf(ss * inv_cs + eps);
    const float rinv = 1.f / rn;
    if (kg == 0) norm[pix] = rn;
    unsigned char* dst = vol + ((v * S) * P + p) * 32 + kg * 8;
    const long dstride = P * 32;
    for (int d = 0; d < S; ++d) {
      const f32x4 acc = mfma16(*(const bf16x4m*)(aw + d * 512), b, (f32x4){0.f, 0.f, 0.f, 0.f});
      const f32x4 bv = *(const f32x4*)(bw + d * 16);
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = __builtin_fmaf(acc[e], he, bv[e]);
        t = fmaxf(t, t * slope);
        o[e] = t * rinv;
      }
      *(bf16x4m*)(dst + d * dstride) = __builtin_convertvector(o, bf16x4m);
    }
  }
}

// wtab_t: bf16 [S][16 cin][16 c], entry = W[c*S + d][cin].  gvol / yvol: bf16 volumes as above; nrm [R]; x fp32 rows [R][16].
// gx: fp32 rows [R][16] (bf16 VALUES when round_gx); pw: [gridDim.x][S][16 c][16 cin] partial weight gradients, pb:
// [gridDim.x][S][16 c] partial bias gradients.  SD = S / 8 depths per wave.
template <int SD>
__global__ void __launch_bounds__(512, 2) lift_bwd_mfma_kernel(const unsigned char* __restrict__ gvol, const unsigned char* __restrict__ yvol,
                                                              const float* __restrict__ nrm, const float* __restrict__ x,
                                                              const __bf16* __restrict__ wtab_t, float* __restrict__ gx,
                                                              float* __restrict__ pw, float* __restrict__ pb, long R, long P, float he,
                                                              float slope, int round_gx) {
  constexpr int S = SD * 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lw = smem;                                       // S * 512 B: [d][cin][c]
  unsigned char* scr = smem + S * 512;                            // per wave: x tile 512 B + gp tile 512 B
  float* red = (float*)(scr + 8 * 1024);                          // [8 waves][16 pixels]
  f32x4* red2 = (f32x4*)(red + 8 * 16);                           // [8 waves][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kg = lane >> 4;
  for (int i = tid; i < S * 32; i += 512) ((f32x4*)lw)[i] = ((const f32x4*)wtab_t)[i];
  __syncthreads();
  unsigned char* xs = scr + wave * 1024;
  unsigned char* gs = xs + 512;
  // own-layout slot of this lane in a 16 x 16 bf16 tile [row n][col 4 kg ..], and its transpose-load address:
  // rows 4 kg + (n >> 2), column quad n & 3 -> the lane receives column n of rows 4 kg .. 4 kg + 3
  const int own = (n * 16 + kg * 4) * 2;
  const int trp = ((kg * 4 + (n >> 2)) * 16 + (n & 3) * 4) * 2;
  f32x4 accw[SD];
  float accb[SD];
#pragma unroll
  for (int j = 0; j < SD; ++j) { accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accb[j] = 0.f; }
  const long groups = R / 16;
  const float inv_cs = 1.f / (16.f * (float)S);
  const long dstride = P * 32;
  for (long g = blockIdx.x; g < groups; g += gridDim.x) {
    const long pix = g * 16 + n;
    const long v = pix / P, p = pix - v * P;
    const long base = ((v * S + wave * SD) * P + p) * 32 + kg * 8;
    f32x4 gg[SD], yy[SD];
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < SD; ++j) {
      gg[j] = __builtin_convertvector(*(const bf16x4m*)(gvol + base + j * dstride), f32x4);
      yy[j] = __builtin_convertvector(*(const bf16x4m*)(yvol + base + j * dstride), f32x4);
    }
    // x^T operand of the weight gradient: rows of x -> LDS tile [pixel][cin] -> transposed back: lane (cin n, pixels 4 kg ..)
    *(bf16x4m*)(xs + own) = __builtin_convertvector(*(const f32x4*)(x + pix * 16 + kg * 4), bf16x4m);
#pragma unroll
    for (int j = 0; j < SD; ++j) q += gg[j][0] * yy[j][0] + gg[j][1] * yy[j][1] + gg[j][2] * yy[j][2] + gg[j][3] * yy[j][3];
    q = quad_sum(q);
    if (kg == 0) red[wave * 16 + n] = q;
    asm volatile("" ::: "memory");
    const bf16x4m xt = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4m*)(xs + trp));
    const float rinv = 1.f / nrm[pix];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    float dot = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) dot += red[w * 16 + n];
    const float dk = dot * inv_cs;
    f32x4 accx = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < SD; ++j) {
      f32x4 gp;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = (gg[j][e] - yy[j][e] * dk) * rinv;
        gp[e] = yy[j][e] > 0.f ? t : t * slope;
      }
      const bf16x4m gpb = __builtin_convertvector(gp, bf16x4m);
      const int d = wave * SD + j;
      // gx[cin][pixel] += W_d^T[cin][c] gp[c][pixel]: A = lane (cin n, c 4 kg ..), B = gp as held (pixel n, c 4 kg ..)
      accx = mfma16(*(const bf16x4m*)(lw + d * 512 + own), gpb, accx);
      // gp -> [pixel][c] tile -> transposed: lane (c n, pixels 4 kg ..) = the A operand of gW_d[c][cin] += gp[c][pixel] x[pixel][cin]
      *(bf16x4m*)(gs + own) = gpb;
      asm volatile("" ::: "memory");                                // (the transpose load below reads what was just written)
      const bf16x4m gt = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4m*)(gs + trp));
      accw[j] = mfma16(gt, xt, accw[j]);
      const f32x4 gf = __builtin_convertvector(gt, f32x4);
      accb[j] += (gf[0] + gf[1]) + (gf[2] + gf[3]);
    }
    red2[wave * 64 + lane] = accx;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave == 0) {
      f32x4 s = red2[lane];
#pragma unroll
      for (int w = 1; w < 8; ++w) s += red2[w * 64 + lane];
      s = s * he;
      if (round_gx) s = __builtin_convertvector(__builtin_convertvector(s, bf16x4m), f32x4);
      *(f32x4*)(gx + pix * 16 + kg * 4) = s;                       // lane (pixel n, cin 4 kg ..)
    }
  }
  // this workgroup's partial sums: gW_d[c = 4 kg + i][cin = n]; bias: channel c = n, summed over the four pixel quads
  float* ow = pw + ((long)blockIdx.x * S + wave * SD) * 256;
  float* ob = pb + ((long)blockIdx.x * S + wave * SD) * 16;
#pragma unroll
  for (int j = 0; j < SD; ++j) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ow[j * 256 + (kg * 4 + e) * 16 + n] = accw[j][e];
    const float b = quad_sum(accb[j]);
    if (kg == 0) ob[j * 16 + n] = b;
  }
}

// gw[(c*S + d)][cin] = he * sum_wg pw[wg][d][c][cin]; gb[c*S + d] = sum_wg pb[wg][d][c] (workgroup order)
__global__ void __launch_bounds__(256) lift_bwd_reduce_kernel(const float* __restrict__ pw, const float* __restrict__ pb, int nwg, int S,
                                                              float he, float* __restrict__ gw, float* __restrict__ gb) {
  const int i = blockIdx.x * 256 + threadIdx.x;                   // over S * 256 weight entries, then S * 16 bias entries
  const int nw = S * 256;
  if (i < nw) {
    float s = 0.f;
    for (int w = 0; w < nwg; ++w) s += pw[(long)w * nw + i];
    const int d = i >> 8, c = (i >> 4) & 15, ci = i & 15;
    gw[(c * S + d) * 16 + ci] = s * he;
  } else if (i < nw + S * 16) {
    const int k = i - nw;
    float s = 0.f;
    for (int w = 0; w < nwg; ++w) s += pb[(long)w * S * 16 + k];
    const int d = k >> 4, c = k & 15;
    gb[c * S + d] = s;
  }
}

// ---- the 3-D -> 2-D factor projection of the training step's renderer (reference modules/geometry.py:731-749
// FactorProjection3d2d: the depth axis folded into the channels, index c*S + d, of a pointwise conv K = 16*S -> 16, LeakyReLU,
// PixelNorm) on the same idea, for 16 volume channels and 16 output channels: the volume is read as it lies (bf16 records),
// one MFMA per depth accumulates a 16-pixel group's 16 outputs; backward: the volume gradient record by record
// (gx_d[c][pixel] = W_d^T gp), the weight gradient per depth with both operands through the LDS transpose load.  Round 5:
// fp32 copy of the volume + conv1x1 forward; conv1x1 into an fp32 volume + rounding pass + row copy of the volume + generic
// weight gradient + cast backward (2.9 ms at 8 views of 128^3; now 0.4).
// wtab: bf16 [S][16 o][16 c] = W[o][c*S + d]; y: fp32 rows [R][16]; norm [R].
__global__ void __launch_bounds__(256, 2) proj16_fwd_kernel(const unsigned char* __restrict__ vol, const __bf16* __restrict__ wtab,
                                                           const float* __restrict__ bias, float* __restrict__ y, float* __restrict__ norm,
                                                           long R, long P, int S, float he, float slope, float eps) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, kg = lane >> 4;
  for (int i = tid; i < S * 32; i += 256) ((f32x4*)smem)[i] = ((const f32x4*)wtab)[i];
  __syncthreads();
  const long groups = R / 16;
  const unsigned char* aw = smem + (n * 16 + kg * 4) * 2;          // A: row o = n, c 4 kg .. (+ d * 512)
  f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) bv = *(const f32x4*)(bias + kg * 4);
  for (long g = (long)blockIdx.x * 4 + wave; g < groups; g += (long)gridDim.x * 4) {
    const long pix = g * 16 + n;
    const long v = pix / P, p = pix - v * P;
    const unsigned char* src = vol + ((v * S) * P + p) * 32 + kg * 8;
    const long dstride = P * 32;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int d = 0; d < S; ++d) acc = mfma16(*(const bf16x4m*)(aw + d * 512), *(const bf16x4m*)(src + d * dstride), acc);
    f32x4 t;
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float u = __builtin_fmaf(acc[e], he, bv[e]);
      u = fmaxf(u, u * slope);
      t[e] = u;
      ss = __builtin_fmaf(u, u, ss);
    }
    ss = quad_sum(ss);
    const float rn = sqrtf(ss * (1.f / 16.f) + eps);
    const float rinv = 1.f / rn;
    *(f32x4*)(y + pix * 16 + kg * 4) = t * rinv;
    if (kg == 0) norm[pix] = rn;
  }
}

// gp: fp32 rows [R][16] (rounded to bf16 here); wtab_t: bf16 [S][16 c][16 o] = W[o][c*S + d]; gxvol: bf16 records;
// pw: [gridDim.x][S][16 o][16 c] partial weight gradients.  SD = S / 8 depths per wave.
template <int SD>
__global__ void __launch_bounds__(512, 2) proj16_bwd_kernel(const float* __restrict__ gp, const unsigned char* __restrict__ vol,
                                                           const __bf16* __restrict__ wtab_t, unsigned char* __restrict__ gxvol,
                                                           float* __restrict__ pw, long R, long P, float he) {
  constexpr int S = SD * 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lw = smem;
  unsigned char* scr = smem + S * 512;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = lane & 15, kg = lane >> 4;
  for (int i = tid; i < S * 32; i += 512) ((f32x4*)lw)[i] = ((const f32x4*)wtab_t)[i];
  __syncthreads();
  unsigned char* gs = scr + wave * 1024;
  unsigned char* xs = gs + 512;
  const int own = (n * 16 + kg * 4) * 2;
  const int trp = ((kg * 4 + (n >> 2)) * 16 + (n & 3) * 4) * 2;
  f32x4 accw[SD];
#pragma unroll
  for (int j = 0; j < SD; ++j) accw[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const long groups = R / 16;
  const long dstride = P * 32;
  for (long g = blockIdx.x; g < groups; g += gridDim.x) {
    const long pix = g * 16 + n;
    const long v = pix / P, p = pix - v * P;
    const long base = ((v * S + wave * SD) * P + p) * 32 + kg * 8;
    const bf16x4m gpb = __builtin_convertvector(*(const f32x4*)(gp + pix * 16 + kg * 4), bf16x4m);    // lane (pixel n, o 4 kg ..)
    bf16x4m xr[SD];
#pragma unroll
    for (int j = 0; j < SD; ++j) xr[j] = *(const bf16x4m*)(vol + base + j * dstride);
    *(bf16x4m*)(gs + own) = gpb;
    asm volatile("" ::: "memory");
    const bf16x4m gt = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4m*)(gs + trp));          // lane (o n, pixels 4 kg ..)
#pragma unroll
    for (int j = 0; j < SD; ++j) {
      const int d = wave * SD + j;
      // gx_d[c][pixel] = W_d^T[c][o] gp[o][pixel]
      const f32x4 o = mfma16(*(const bf16x4m*)(lw + d * 512 + own), gpb, (f32x4){0.f, 0.f, 0.f, 0.f}) * he;
      *(bf16x4m*)(gxvol + base + j * dstride) = __builtin_convertvector(o, bf16x4m);
      // gW_d[o][c] += gp[o][pixel] x_d[pixel][c]
      *(bf16x4m*)(xs + own) = xr[j];
      asm volatile("" ::: "memory");
      const bf16x4m xt = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4m*)(xs + trp));        // lane (c n, pixels 4 kg ..)
      accw[j] = mfma16(gt, xt, accw[j]);
    }
  }
  float* ow = pw + ((long)blockIdx.x * S + wave * SD) * 256;
#pragma unroll
  for (int j = 0; j < SD; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) ow[j * 256 + (kg * 4 + e) * 16 + n] = accw[j][e];
}

// gw[o][c*S + d] = he * sum_wg pw[wg][d][o][c]
__global__ void __launch_bounds__(256) proj16_bwd_reduce_kernel(const float* __restrict__ pw, int nwg, int S, float he, float* __restrict__ gw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int nw = S * 256;
  if (i >= nw) return;
  float s = 0.f;
  for (int w = 0; w < nwg; ++w) s += pw[(long)w * nw + i];
  const int d = i >> 8, o = (i >> 4) & 15, c = i & 15;
  gw[o * (16 * S) + c * S + d] = s * he;
}

int lift_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, v = 0;
    cus = (hipGetDevice(&dev) == hipSuccess &&
           hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cus;
}

}  // namespace

extern "C" int lf_lift16_fwd(const float* x, const void* wtab, const float* btab, void* vol, float* norm, long R, long P, int S,
                             float he, float slope, float eps, void* stream) {
  lf_clear_error();
  if (R <= 0 || P <= 0 || S <= 0 || (R % 16) || (P % 16) || (R % P) || S > 256 || x == nullptr || wtab == nullptr || vol == nullptr ||
      norm == nullptr) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(wtab) || !lf_aligned16(vol) || (btab && !lf_aligned16(btab))) return LF_EALIGN;
  const size_t shmem = (size_t)S * 512 + (size_t)S * 64;
  static lf_devmask_t attr;
  if (shmem > 48 * 1024) {
    hipError_t e = lf_ensure_dyn_lds(attr, (const void*)lift_fwd_mfma_kernel, 150 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  const long groups = R / 16;
  long nb = 2L * lift_cus();
  if (nb * 4 > groups) nb = (groups + 3) / 4;
  hipLaunchKernelGGL(lift_fwd_mfma_kernel, dim3((unsigned)nb), dim3(256), shmem, (hipStream_t)stream, x, (const __bf16*)wtab, btab,
                     (unsigned char*)vol, norm, R, P, S, he, slope, eps);
  return lf_launch_status();
}

extern "C" size_t lf_lift16_bwd_scratch_bytes(int S) { return (size_t)lift_cus() * (size_t)S * (256 + 16) * sizeof(float); }

extern "C" int lf_lift16_bwd(const void* gvol, const void* yvol, const float* norm, const float* x, const void* wtab_t, float* gx, float* gw,
                             float* gb, void* scratch, size_t scratch_bytes, long R, long P, int S, float he, float slope, int round_gx,
                             void* stream) {
  lf_clear_error();
  if (R <= 0 || P <= 0 || (R % 16) || (P % 16) || (R % P) || gvol == nullptr || yvol == nullptr || norm == nullptr || x == nullptr ||
      wtab_t == nullptr || gx == nullptr || gw == nullptr || gb == nullptr) return LF_EINVAL;
  if (S != 16 && S != 32 && S != 64 && S != 128) return LF_EINVAL;
  if (!lf_aligned16(gvol) || !lf_aligned16(yvol) || !lf_aligned16(x) || !lf_aligned16(wtab_t) || !lf_aligned16(gx) || !lf_aligned16(scratch)) return LF_EALIGN;
  const long groups = R / 16;
  int nwg = lift_cus();
  if (nwg > groups) nwg = (int)groups;
  if (scratch == nullptr || scratch_bytes < (size_t)nwg * S * (256 + 16) * sizeof(float)) return LF_ENOSPC;
  float* pw = (float*)scratch;
  float* pb = pw + (size_t)nwg * S * 256;
  const size_t shmem = (size_t)S * 512 + 8 * 1024 + 8 * 16 * 4 + 8 * 64 * 16;
  hipStream_t s = (hipStream_t)stream;
  typedef void (*kern_t)(const unsigned char*, const unsigned char*, const float*, const float*, const __bf16*, float*, float*, float*, long,
                         long, float, float, int);
  kern_t k = S == 128 ? lift_bwd_mfma_kernel<16> : (S == 64 ? lift_bwd_mfma_kernel<8> : (S == 32 ? lift_bwd_mfma_kernel<4> : lift_bwd_mfma_kernel<2>));
  static lf_devmask_t attr[4];
  if (shmem > 48 * 1024) {
    hipError_t e = lf_ensure_dyn_lds(attr[S == 128 ? 0 : (S == 64 ? 1 : (S == 32 ? 2 : 3))], (const void*)k, 150 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(512), shmem, s, (const unsigned char*)gvol, (const unsigned char*)yvol, norm, x,
                     (const __bf16*)wtab_t, gx, pw, pb, R, P, he, slope, round_gx);
  int st = lf_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(lift_bwd_reduce_kernel, dim3((unsigned)((S * 272 + 255) / 256)), dim3(256), 0, s, (const float*)pw, (const float*)pb, nwg,
                     S, he, gw, gb);
  return lf_launch_status();
}

extern "C" int lf_proj16_fwd(const void* vol, const void* wtab, const float* bias, float* y, float* norm, long R, long P, int S, float he,
                             float slope, float eps, void* stream) {
  lf_clear_error();
  if (R <= 0 || P <= 0 || S <= 0 || S > 256 || (R % 16) || (P % 16) || (R % P) || vol == nullptr || wtab == nullptr || y == nullptr ||
      norm == nullptr) return LF_EINVAL;
  if (!lf_aligned16(vol) || !lf_aligned16(wtab) || !lf_aligned16(y) || (bias && !lf_aligned16(bias))) return LF_EALIGN;
  const size_t shmem = (size_t)S * 512;
  static lf_devmask_t attr;
  if (shmem > 48 * 1024) {
    hipError_t e = lf_ensure_dyn_lds(attr, (const void*)proj16_fwd_kernel, 150 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  const long groups = R / 16;
  long nb = 2L * lift_cus();
  if (nb * 4 > groups) nb = (groups + 3) / 4;
  hipLaunchKernelGGL(proj16_fwd_kernel, dim3((unsigned)nb), dim3(256), shmem, (hipStream_t)stream, (const unsigned char*)vol,
                     (const __bf16*)wtab, bias, y, norm, R, P, S, he, slope, eps);
  return lf_launch_status();
}

extern "C" size_t lf_proj16_bwd_scratch_bytes(int S) { return (size_t)lift_cus() * (size_t)S * 256 * sizeof(float); }

extern "C" int lf_proj16_bwd(const float* gp, const void* vol, const void* wtab_t, void* gxvol, float* gw, void* scratch,
                             size_t scratch_bytes, long R, long P, int S, float he, void* stream) {
  lf_clear_error();
  if (R <= 0 || P <= 0 || (R % 16) || (P % 16) || (R % P) || gp == nullptr || vol == nullptr || wtab_t == nullptr || gxvol == nullptr ||
      gw == nullptr) return LF_EINVAL;
  if (S != 16 && S != 32 && S != 64 && S != 128) return LF_EINVAL;
  if (!lf_aligned16(gp) || !lf_aligned16(vol) || !lf_aligned16(wtab_t) || !lf_aligned16(gxvol) || !lf_aligned16(scratch)) return LF_EALIGN;
  const long groups = R / 16;
  int nwg = lift_cus();
  if (nwg > groups) nwg = (int)groups;
  if (scratch == nullptr || scratch_bytes < (size_t)nwg * S * 256 * sizeof(float)) return LF_ENOSPC;
  const size_t shmem = (size_t)S * 512 + 8 * 1024;
  hipStream_t s = (hipStream_t)stream;
  typedef void (*kern_t)(const float*, const unsigned char*, const __bf16*, unsigned char*, float*, long, long, float);
  kern_t k = S == 128 ? proj16_bwd_kernel<16> : (S == 64 ? proj16_bwd_kernel<8> : (S == 32 ? proj16_bwd_kernel<4> : proj16_bwd_kernel<2>));
  static lf_devmask_t attr[4];
  if (shmem > 48 * 1024) {
    hipError_t e = lf_ensure_dyn_lds(attr[S == 128 ? 0 : (S == 64 ? 1 : (S == 32 ? 2 : 3))], (const void*)k, 150 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(512), shmem, s, gp, (const unsigned char*)vol, (const __bf16*)wtab_t, (unsigned char*)gxvol,
                     (float*)scratch, R, P, he);
  int st = lf_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(proj16_bwd_reduce_kernel, dim3((unsigned)((S * 256 + 255) / 256)), dim3(256), 0, s, (const float*)scratch, nwg, S, he, gw);
  return lf_launch_status();
}
