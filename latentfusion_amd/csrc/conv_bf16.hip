// bf16-autocast form of the fused 16 -> 16 channel conv3d block step for the TRAINING step (BASELINE cfg 5):
//   y = PixelNorm(LeakyReLU(bf16(bf16(conv3d(bf16(x), bf16(W))) * he) + b))
// = what torch autocast makes of Equalized.forward + LeakyReLU + PixelNorm (latentfusion/modules/equalized.py:57-64,
// blocks.py:152-158 under recon/models.py:199,405 `autocast(enabled=self.training)`): the convolution runs on
// half-precision operands with fp32 accumulation and returns a half tensor, the He scale is applied in half, and the
// fp32 bias promotes everything after it back to fp32.  Operands are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) while the
// halo is staged, products run on v_mfma_f32_16x16x16_bf16 -- DIRECT convolution, one MFMA per tap and 16 voxels (a
// Winograd transform of bf16 data is not bf16-exact, so the minimal-filtering kernels cannot reproduce autocast).
// The same kernel with transposed / flipped weights is the data gradient (flags = 0, no bias; result rounded to bf16
// like autocast's conv backward).
//
// Tile 4 x 8 x 16 output voxels per 256-thread workgroup; the 6 x 10 x 18 halo lives in LDS as bf16 (32 B per voxel,
// 34.5 KB), the four 8-byte channel quarters of a voxel swizzled by bit 3 of its x so that the b64 operand reads of
// 16 consecutive voxels x 2 quarters spread over all banks.  Wave w owns z plane w: 8 rows of 16 voxels = 8 independent
// accumulators, 27 weight fragments (54 VGPRs) resident.
#include "lf_common.h"

namespace {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int BZ = 4, BY = 8, BX = 16, GZ = BZ + 2, GY = BY + 2, GX = BX + 2;
constexpr int GHALO = GZ * GY * GX;                              // 1080 voxels
constexpr int BF_LDS = GHALO * 32;                               // 34,560 B

__device__ __forceinline__ int halo_slot(int vox, int lx, int q) { return vox * 32 + ((q ^ (((lx >> 3) & 1) << 1)) << 3); }

__device__ __forceinline__ float round_bf16(float v) { return (float)(__bf16)v; }

__global__ void __launch_bounds__(256) round_bf16_kernel(const f32x4* __restrict__ x, f32x4* __restrict__ y, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = x[i];
  y[i] = __builtin_convertvector(__builtin_convertvector(v, bf16x4), f32x4);
}

__global__ void __launch_bounds__(256) round_bf16_scalar_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = round_bf16(x[i]);
}

__global__ void __launch_bounds__(256) conv3d_c16_bf16_kernel(
    const float* __restrict__ x, const s16x4* __restrict__ wpack, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ norm_out, int D, int H, int W, int tiles_x, int tiles_y, int tiles_z, float he, unsigned flags,
    float slope, float eps, int round_out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[BF_LDS];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, kg = lane >> 4;
  int t = blockIdx.x;
  const int bx = t % tiles_x; t /= tiles_x;
  const int by = t % tiles_y; t /= tiles_y;
  const int bz = t % tiles_z;
  const int smp = t / tiles_z;
  const long nvox = (long)D * H * W;
  const float* xs = x + (long)smp * nvox * 16;
  const int ox = bx * BX - 1, oy = by * BY - 1, oz = bz * BZ - 1;

  // ---- stage the halo: fp32 global -> bf16 LDS (zero padding outside the volume).  All of a thread's loads are
  // issued before the first conversion (one exposed memory latency per tile instead of seventeen) ----
  constexpr int NST = (GHALO * 4 + 255) / 256;                   // 17 float4 pieces per thread
  f32x4 st[NST];
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int i = it * 256 + tid;
    const int vox = i >> 2, q = i & 3;
    const int lx = vox % GX, ly = (vox / GX) % GY, lz = vox / (GX * GY);
    const int gx = ox + lx, gy = oy + ly, gz = oz + lz;
    st[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < GHALO * 4 && (unsigned)gx < (unsigned)W && (unsigned)gy < (unsigned)H && (unsigned)gz < (unsigned)D)
      st[it] = *(const f32x4*)(xs + (((long)gz * H + gy) * W + gx) * 16 + q * 4);
  }
#pragma unroll
  for (int it = 0; it < NST; ++it) {
    const int i = it * 256 + tid;
    const int vox = i >> 2, q = i & 3;
    if (i < GHALO * 4) *(bf16x4*)(lds + halo_slot(vox, vox % GX, q)) = __builtin_convertvector(st[it], bf16x4);
  }
  // ---- weights: A operand of tap t, lane (cout = n, cin group kg) ----
  s16x4 wreg[27];
#pragma unroll
  for (int tap = 0; tap < 27; ++tap) wreg[tap] = wpack[tap * 64 + lane];
  int offx[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) offx[dx] = halo_slot(n + dx, n + dx, kg);        // (row base added per tap)
  __syncthreads();

  f32x4 acc[BY];
#pragma unroll
  for (int r = 0; r < BY; ++r) acc[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int tap = (dz * 3 + dy) * 3 + dx;
#pragma unroll
        for (int r = 0; r < BY; ++r) {
          const int rowbase = (((w + dz) * GY + (r + dy)) * GX) * 32;           // wave-uniform
          const s16x4 b = *(const s16x4*)(lds + rowbase + offx[dx]);
          acc[r] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wreg[tap], b, acc[r], 0, 0, 0);
        }
      }

  // ---- epilogue: lane holds couts kg*4 .. +3 of voxel (gz, gy = by*8 + r, gx = bx*16 + n) ----
  const int gz = bz * BZ + w, gx = bx * BX + n;
  f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (bias != nullptr) bv = *(const f32x4*)(bias + kg * 4);
  float* ys = y + (long)smp * nvox * 16;
  float* ns = norm_out ? norm_out + (long)smp * nvox : nullptr;
#pragma unroll
  for (int r = 0; r < BY; ++r) {
    const int gy = by * BY + r;
    f32x4 v;
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // autocast: the convolution returns a half tensor, `* he` stays in half, `+ bias` (fp32) promotes to fp32
      float u = round_out ? round_bf16(round_bf16(acc[r][e]) * he) : acc[r][e] * he;
      if (round_out == 2) u = round_bf16(u);                   // data gradient: the result itself is a half tensor
      u += bv[e];
      if (flags & LF_EPI_LRELU) u = fmaxf(u, u * slope);
      v[e] = u;
      ss += u * u;
    }
    float rn = 1.f;
    if (flags & LF_EPI_PIXELNORM) {
      ss += __shfl_xor(ss, 16, 64);                              // the 16 channels of a voxel sit in lanes n, n+16, n+32, n+48
      ss += __shfl_xor(ss, 32, 64);
      rn = sqrtf(ss * (1.f / 16.f) + eps);
      const float rinv = 1.f / rn;
      v *= rinv;
    }
    if (gx < W && gy < H && gz < D) {
      const long vox = ((long)gz * H + gy) * W + gx;
      *(f32x4*)(ys + vox * 16 + kg * 4) = v;
      if ((flags & LF_EPI_PIXELNORM) && ns != nullptr && kg == 0) ns[vox] = rn;
    }
  }
}

}  // namespace

// y = round-to-bf16(x), kept in fp32 containers (the autocast cast of an operand)
extern "C" int lf_round_bf16(const float* x, float* y, long n, void* stream) {
  lf_clear_error();
  if (n <= 0) return LF_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  if ((n & 3) == 0 && lf_aligned16(x) && lf_aligned16(y))
    hipLaunchKernelGGL(round_bf16_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const f32x4*)x, (f32x4*)y, n / 4);
  else
    hipLaunchKernelGGL(round_bf16_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, y, n);
  return lf_launch_status();
}

// bf16 elements of the weight pack: [27 taps][64 lanes][4]
extern "C" size_t lf_conv3d_c16_bf16_wpack_elems(void) { return (size_t)27 * 64 * 4; }

extern "C" int lf_conv3d_c16_bf16(const float* x, const void* wpack, const float* bias, float* y, float* norm_out, int N, int D,
                                  int H, int W, float he, unsigned flags, float slope, float eps, int round_out, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || round_out < 0 || round_out > 2) return LF_EINVAL;
  if (flags & ~(LF_EPI_LRELU | LF_EPI_PIXELNORM)) return LF_EINVAL;
  if (!lf_aligned16(x) || !lf_aligned16(y) || (((uintptr_t)wpack) & 7u) || (bias && !lf_aligned16(bias))) return LF_EALIGN;
  const int tx = (W + BX - 1) / BX, ty = (H + BY - 1) / BY, tz = (D + BZ - 1) / BZ;
  const long nt = (long)N * tx * ty * tz;
  if (nt > 0x7fffffffL) return LF_EINVAL;
  hipLaunchKernelGGL(conv3d_c16_bf16_kernel, dim3((unsigned)nt), dim3(256), 0, (hipStream_t)stream, x, (const s16x4*)wpack, bias, y,
                     norm_out, D, H, W, tx, ty, tz, he, flags, slope, eps, round_out);
  return lf_launch_status();
}
