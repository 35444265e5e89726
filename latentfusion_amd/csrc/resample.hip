// Camera <-> object voxel resampling for gfx950 (trilinear gather / coefficient-gradient
// reduction / splat).  Replaces F.grid_sample(…, padding_mode='border', align_corners=False) of
//   ObjectToCameraTransform.forward  latentfusion/modules/geometry.py:669-690
//   CameraToObjectTransform.forward  latentfusion/modules/geometry.py:625-657
// The sampling grid is evaluated per voxel from a per-sample coefficient block (see lf_hip.h);
// volumes are channels-last so that every trilinear tap is one contiguous C-float record.
#include "lf_common.h"
#include <type_traits>

namespace {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct Tap {
  int x0, y0, z0, x1, y1, z1;   // clamped integer corners
  float tx, ty, tz;             // fractional offsets
  float mx, my, mz;             // d(ix)/d(gx) incl. border-clip mask (0 outside the volume)
};

// i-th point of torch.linspace(0, 1, size): start + step*i in the first half, end - step*(size-1-i) in the
// second (ATen's symmetric formula), step = 1/(size-1) computed once on the host
__device__ __forceinline__ float lattice01(int i, int size, float step) {
  return (i < size / 2) ? step * (float)i : 1.f - step * (float)(size - 1 - i);
}

struct Steps { float w, h, d; };                              // 1/(W-1), 1/(H-1), 1/(D-1) (0 for size 1)

template <int KIND>
__device__ __forceinline__ void eval_grid(const float* __restrict__ cf, int x, int y, int z,
                                          int W, int H, int D, Steps st, float& gx, float& gy, float& gz,
                                          float& a, float& b, float& k) {
  // lattice coordinates in [0,1] (torch.linspace(0,1,S): geometry.py:476-480)
  a = lattice01(x, W, st.w);
  b = lattice01(y, H, st.h);
  k = lattice01(z, D, st.d);
  if (KIND == LF_MAP_O2C) {
    const float ak = a * k, bk = b * k;
    gx = cf[0] + cf[3] * a + cf[6] * b + cf[9] * k + cf[12] * ak + cf[15] * bk;
    gy = cf[1] + cf[4] * a + cf[7] * b + cf[10] * k + cf[13] * ak + cf[16] * bk;
    gz = cf[2] + cf[5] * a + cf[8] * b + cf[11] * k + cf[14] * ak + cf[17] * bk;
  } else {
    // lattice in [-1,1] (torch.linspace(-c/2,c/2,S) / (c/2): geometry.py:599-611)
    const float lx = 2.f * a - 1.f, ly = 2.f * b - 1.f, lz = 2.f * k - 1.f;
    const float n0 = cf[0] * lx + cf[1] * ly + cf[2] * lz + cf[3];
    const float n1 = cf[4] * lx + cf[5] * ly + cf[6] * lz + cf[7];
    const float n2 = cf[8] * lx + cf[9] * ly + cf[10] * lz + cf[11];
    const float dn = cf[12] * lx + cf[13] * ly + cf[14] * lz + cf[15];
    gx = n0 / dn;
    gy = n1 / dn;
    gz = n2;
  }
}

__device__ __forceinline__ void unnormalize_clip(float g, int size, float& pos, float& mult) {
  // grid_sampler_unnormalize (align_corners=False) + clip_coordinates_set_grad (border)
  float p = ((g + 1.f) * (float)size - 1.f) * 0.5f;
  const float hi = (float)(size - 1);
  mult = (p > 0.f && p < hi) ? 0.5f * (float)size : 0.f;
  p = fminf(fmaxf(p, 0.f), hi);
  pos = p;
}

__device__ __forceinline__ Tap make_tap(float gx, float gy, float gz, int W, int H, int D) {
  Tap t;
  float px, py, pz;
  unnormalize_clip(gx, W, px, t.mx);
  unnormalize_clip(gy, H, py, t.my);
  unnormalize_clip(gz, D, pz, t.mz);
  // NaN coordinates (degenerate cameras) sample voxel 0, like ATen's clip of NaN
  if (!(px == px)) px = 0.f;
  if (!(py == py)) py = 0.f;
  if (!(pz == pz)) pz = 0.f;
  const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
  t.tx = px - fx; t.ty = py - fy; t.tz = pz - fz;
  t.x0 = (int)fx; t.y0 = (int)fy; t.z0 = (int)fz;
  t.x1 = min(t.x0 + 1, W - 1); t.y1 = min(t.y0 + 1, H - 1); t.z1 = min(t.z0 + 1, D - 1);
  return t;
}

// Workgroups are dispatched round-robin over the 8 XCDs (one L2 each): flat workgroup id b runs on XCD b % 8.
// Giving XCD k the k-th contiguous eighth of the work list keeps neighbouring tiles -- whose gather footprints
// overlap -- behind one L2 instead of eight.  (Measured neutral for these kernels: the gathered volume lives in the
// 256 MB Infinity Cache either way; kept for the coefficient gradient, whose blocks are large.)
__device__ __forceinline__ unsigned xcd_contiguous(unsigned b, unsigned nb) {
  return (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
}

// VEC = 4: one thread per (voxel, 4-channel group), C % 4 == 0.  VEC = 1: one thread per (voxel, channel).
// A block covers a compact 2^lx x 2^ly x 2^lz output tile (4x4x4 for C = 16), so that the 8-corner
// footprints of its voxels overlap in L1; 32-bit index math only.
template <int KIND, int VEC>
__global__ void __launch_bounds__(256) resample_fwd_kernel(
    const float* __restrict__ vol, long vol_bstride, const float* __restrict__ coef,
    float* __restrict__ out, int N, int D, int H, int W, int C, int lpt, int lx, int ly, int lz, int nbz, Steps st) {
  // lpt = threads cooperating on one voxel (<= 256); tile = (1<<lx, 1<<ly, 1<<lz) voxels per block
  const int lpv = C / VEC;
  const int n = blockIdx.z / nbz, bz = blockIdx.z - n * nbz;
  const int slot = threadIdx.x / lpt, q0 = threadIdx.x - slot * lpt;
  if (slot >= (1 << (lx + ly + lz))) return;
  const int x = (blockIdx.x << lx) + (slot & ((1 << lx) - 1));
  const int y = (blockIdx.y << ly) + ((slot >> lx) & ((1 << ly) - 1));
  const int z = (bz << lz) + (slot >> (lx + ly));
  if (x >= W || y >= H || z >= D) return;
  const float* cf = coef + (long)n * LF_MAP_COEFS;
  float gx, gy, gz, a, b, k;
  eval_grid<KIND>(cf, x, y, z, W, H, D, st, gx, gy, gz, a, b, k);
  const Tap t = make_tap(gx, gy, gz, W, H, D);
  const float wx1 = t.tx, wx0 = 1.f - t.tx, wy1 = t.ty, wy0 = 1.f - t.ty, wz1 = t.tz, wz0 = 1.f - t.tz;
  const float w000 = wx0 * wy0 * wz0, w001 = wx1 * wy0 * wz0, w010 = wx0 * wy1 * wz0, w011 = wx1 * wy1 * wz0;
  const float w100 = wx0 * wy0 * wz1, w101 = wx1 * wy0 * wz1, w110 = wx0 * wy1 * wz1, w111 = wx1 * wy1 * wz1;
  const int r00 = (t.z0 * H + t.y0) * W, r01 = (t.z0 * H + t.y1) * W;        // voxel indices of the 4 rows
  const int r10 = (t.z1 * H + t.y0) * W, r11 = (t.z1 * H + t.y1) * W;
  const float* base = vol + (long)n * vol_bstride;
  float* orow = out + ((long)n * D * H * W + ((long)z * H + y) * W + x) * C;
  for (int q = q0; q < lpv; q += lpt) {
    const int co = q * VEC;
    if (VEC == 4) {
      const f32x4 v000 = *(const f32x4*)(base + (long)(r00 + t.x0) * C + co), v001 = *(const f32x4*)(base + (long)(r00 + t.x1) * C + co);
      const f32x4 v010 = *(const f32x4*)(base + (long)(r01 + t.x0) * C + co), v011 = *(const f32x4*)(base + (long)(r01 + t.x1) * C + co);
      const f32x4 v100 = *(const f32x4*)(base + (long)(r10 + t.x0) * C + co), v101 = *(const f32x4*)(base + (long)(r10 + t.x1) * C + co);
      const f32x4 v110 = *(const f32x4*)(base + (long)(r11 + t.x0) * C + co), v111 = *(const f32x4*)(base + (long)(r11 + t.x1) * C + co);
      const f32x4 r = v000 * w000 + v001 * w001 + v010 * w010 + v011 * w011 + v100 * w100 + v101 * w101 + v110 * w110 +
                      v111 * w111;
      __builtin_nontemporal_store(r, (f32x4*)(orow + co));     // streamed output: keep L2 for the gathered volume
    } else {
      const float r = base[(long)(r00 + t.x0) * C + co] * w000 + base[(long)(r00 + t.x1) * C + co] * w001 +
                      base[(long)(r01 + t.x0) * C + co] * w010 + base[(long)(r01 + t.x1) * C + co] * w011 +
                      base[(long)(r10 + t.x0) * C + co] * w100 + base[(long)(r10 + t.x1) * C + co] * w101 +
                      base[(long)(r11 + t.x0) * C + co] * w110 + base[(long)(r11 + t.x1) * C + co] * w111;
      orow[co] = r;
    }
  }
}

// ---- backward w.r.t. the O2C coefficient block ------------------------------------------------
// stage 1: every block reduces a contiguous run of voxels of one sample to 18 partial sums.
// voxels per block: 4096 for big volumes, down to 64 so that small ones (16^3 x 256 channels) still
// launch a few hundred blocks.  A pure function of the shapes -> the reduction order is reproducible.
static int bwd_vox_per_block(long nvox, int N) {
  long v = nvox * N / 2048;
  int p = 64;
  while (p < 4096 && p * 2 <= v) p <<= 1;
  return p;
}
// the block's voxels form a compact tile of 2^bx x 2^by x 2^bz voxels (4096 -> 16^3 ... 64 -> 4^3) walked in
// 4x4x4 sub-tiles, so the 8-corner footprints overlap in L1 / L2 instead of spanning a whole plane
struct BwdTile { int lx, ly, lz, ntx, nty, ntz; };
static BwdTile bwd_tile(int vpb, int D, int H, int W) {
  int lg = 0;
  while ((1 << lg) < vpb) ++lg;                               // vpb = 2^lg, 6 <= lg <= 12
  BwdTile t;
  t.lx = (lg + 2) / 3; t.ly = (lg + 1) / 3; t.lz = lg / 3;
  t.ntx = (W + (1 << t.lx) - 1) >> t.lx; t.nty = (H + (1 << t.ly) - 1) >> t.ly; t.ntz = (D + (1 << t.lz) - 1) >> t.lz;
  return t;
}

template <int VEC>
__global__ void __launch_bounds__(256) resample_bwd_coef_kernel(
    const float* __restrict__ gout, const float* __restrict__ vol, long vol_bstride,
    const float* __restrict__ coef, float* __restrict__ partial, int nblk, int vpb, BwdTile bt,
    int N, int D, int H, int W, int C, int lpv, Steps st) {
  // lpv = lanes cooperating on one voxel: a power of two <= 64 with lpv * VEC >= C
  const unsigned fb = xcd_contiguous(blockIdx.x, gridDim.x);
  const int n = fb / nblk, blk = fb - n * nblk;
  const float* cf = coef + (long)n * LF_MAP_COEFS;
  const int nvox = D * H * W;                                  // < 2^31 (checked by the launcher)
  const int tx = blk % bt.ntx, ty = (blk / bt.ntx) % bt.nty, tz = blk / (bt.ntx * bt.nty);
  const int q = threadIdx.x % lpv;
  const int vslot = threadIdx.x / lpv;
  const int vstep = blockDim.x / lpv;
  float acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = 0.f;
  // all lanes of a wave iterate the same number of times so the shuffles below are convergent
  const int iters = (vpb + vstep - 1) / vstep;
  for (int it = 0; it < iters; ++it) {
    // i-th voxel of the tile: low 6 bits = position inside a 4x4x4 sub-tile, the rest = sub-tile, x fastest
    const int i = it * vstep + vslot;
    const int sub = i >> 6;
    const int sbx = bt.lx - 2, sby = bt.ly - 2;                 // log2 sub-tiles along x, y
    const int x = (tx << bt.lx) + ((sub & ((1 << sbx) - 1)) << 2) + (i & 3);
    const int y = (ty << bt.ly) + (((sub >> sbx) & ((1 << sby) - 1)) << 2) + ((i >> 2) & 3);
    const int z = (tz << bt.lz) + ((sub >> (sbx + sby)) << 2) + ((i >> 4) & 3);
    const bool live = i < vpb && x < W && y < H && z < D;
    const int v = (z * H + y) * W + x;
    float hx = 0.f, hy = 0.f, hz = 0.f, a = 0.f, b = 0.f, k = 0.f;
    if (live && q * VEC < C) {
      float gx, gy, gz;
      eval_grid<LF_MAP_O2C>(cf, x, y, z, W, H, D, st, gx, gy, gz, a, b, k);
      const Tap t = make_tap(gx, gy, gz, W, H, D);
      const float* base = vol + (long)n * vol_bstride + (long)q * VEC;
      const float* g = gout + (((long)n * nvox + v) * C) + (long)q * VEC;
      const long o00 = (long)((t.z0 * H + t.y0) * W) * C, o01 = (long)((t.z0 * H + t.y1) * W) * C;
      const long o10 = (long)((t.z1 * H + t.y0) * W) * C, o11 = (long)((t.z1 * H + t.y1) * W) * C;
      const long x0 = (long)t.x0 * C, x1 = (long)t.x1 * C;
      const float wx1 = t.tx, wx0 = 1.f - t.tx, wy1 = t.ty, wy0 = 1.f - t.ty, wz1 = t.tz, wz0 = 1.f - t.tz;
      float dx = 0.f, dy = 0.f, dz = 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float go = __builtin_nontemporal_load(g + e);     // streamed once: keep L2 for the gathered volume
        const float v000 = base[o00 + x0 + e], v001 = base[o00 + x1 + e];
        const float v010 = base[o01 + x0 + e], v011 = base[o01 + x1 + e];
        const float v100 = base[o10 + x0 + e], v101 = base[o10 + x1 + e];
        const float v110 = base[o11 + x0 + e], v111 = base[o11 + x1 + e];
        dx += go * ((v001 - v000) * (wy0 * wz0) + (v011 - v010) * (wy1 * wz0) +
                    (v101 - v100) * (wy0 * wz1) + (v111 - v110) * (wy1 * wz1));
        dy += go * ((v010 - v000) * (wx0 * wz0) + (v011 - v001) * (wx1 * wz0) +
                    (v110 - v100) * (wx0 * wz1) + (v111 - v101) * (wx1 * wz1));
        dz += go * ((v100 - v000) * (wx0 * wy0) + (v101 - v001) * (wx1 * wy0) +
                    (v110 - v010) * (wx0 * wy1) + (v111 - v011) * (wx1 * wy1));
      }
      hx = dx * t.mx; hy = dy * t.my; hz = dz * t.mz;
    }
    // sum the channel-group lanes of this voxel (xor butterfly stays inside the lpv-lane group)
    for (int o = lpv >> 1; o > 0; o >>= 1) {
      hx += __shfl_xor(hx, o, 64);
      hy += __shfl_xor(hy, o, 64);
      hz += __shfl_xor(hz, o, 64);
    }
    if (q == 0 && live) {
      const float ak = a * k, bk = b * k;
      acc[0] += hx;       acc[1] += hy;       acc[2] += hz;
      acc[3] += hx * a;   acc[4] += hy * a;   acc[5] += hz * a;
      acc[6] += hx * b;   acc[7] += hy * b;   acc[8] += hz * b;
      acc[9] += hx * k;   acc[10] += hy * k;  acc[11] += hz * k;
      acc[12] += hx * ak; acc[13] += hy * ak; acc[14] += hz * ak;
      acc[15] += hx * bk; acc[16] += hy * bk; acc[17] += hz * bk;
    }
  }
  __shared__ float red[4][18];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    const float s = lf_wave_sum(acc[i]);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 18) {
    const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    partial[((long)n * nblk + blk) * 18 + threadIdx.x] = s;
  }
}

// stage 2: fixed-order fp64 reduction of the per-block partials -> gcoef[n][18]
__global__ void __launch_bounds__(256) resample_bwd_coef_reduce(const float* __restrict__ partial,
                                                                int nblk, float* __restrict__ gcoef) {
  const int n = blockIdx.x;
  __shared__ double red[256];
  const int comp = threadIdx.x % 18;        // threads 0..251 -> 14 strided groups of 18
  const int grp = threadIdx.x / 18;
  double s = 0.0;
  if (grp < 14)
    for (int b = grp; b < nblk; b += 14) s += (double)partial[((long)n * nblk + b) * 18 + comp];
  red[threadIdx.x] = (grp < 14) ? s : 0.0;
  __syncthreads();
  if (threadIdx.x < 18) {
    double tot = 0.0;
    for (int g = 0; g < 14; ++g) tot += red[g * 18 + threadIdx.x];
    gcoef[(long)n * 18 + threadIdx.x] = (float)tot;
  }
}

// ---- backward w.r.t. the sampled volume (splat) -----------------------------------------------
template <int KIND>
__global__ void __launch_bounds__(256) resample_bwd_vol_kernel(
    const float* __restrict__ gout, const float* __restrict__ coef, float* __restrict__ gvol,
    long gvol_bstride, int N, int D, int H, int W, int C, Steps st) {
  const long per_sample = (long)D * H * W * C;
  const int n = blockIdx.y;
  const float* cf = coef + (long)n * LF_MAP_COEFS;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < per_sample;
       idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long v = idx / C;
    const int x = (int)(v % W); v /= W;
    const int y = (int)(v % H);
    const int z = (int)(v / H);
    float gx, gy, gz, a, b, k;
    eval_grid<KIND>(cf, x, y, z, W, H, D, st, gx, gy, gz, a, b, k);
    const Tap t = make_tap(gx, gy, gz, W, H, D);
    const float go = gout[(long)n * per_sample + idx];
    float* base = gvol + (long)n * gvol_bstride + c;
    const long sW = C, sH = (long)W * C, sD = (long)H * W * C;
    const float wx1 = t.tx, wx0 = 1.f - t.tx, wy1 = t.ty, wy0 = 1.f - t.ty, wz1 = t.tz, wz0 = 1.f - t.tz;
    atomicAdd(base + t.z0 * sD + t.y0 * sH + t.x0 * sW, go * (wx0 * wy0 * wz0));
    atomicAdd(base + t.z0 * sD + t.y0 * sH + t.x1 * sW, go * (wx1 * wy0 * wz0));
    atomicAdd(base + t.z0 * sD + t.y1 * sH + t.x0 * sW, go * (wx0 * wy1 * wz0));
    atomicAdd(base + t.z0 * sD + t.y1 * sH + t.x1 * sW, go * (wx1 * wy1 * wz0));
    atomicAdd(base + t.z1 * sD + t.y0 * sH + t.x0 * sW, go * (wx0 * wy0 * wz1));
    atomicAdd(base + t.z1 * sD + t.y0 * sH + t.x1 * sW, go * (wx1 * wy0 * wz1));
    atomicAdd(base + t.z1 * sD + t.y1 * sH + t.x0 * sW, go * (wx0 * wy1 * wz1));
    atomicAdd(base + t.z1 * sD + t.y1 * sH + t.x1 * sW, go * (wx1 * wy1 * wz1));
  }
}


// ================================================================================================================
// Lean variants (default): the same arithmetic for the sampling position, but
//   * every address is a 32-bit byte offset into a buffer resource of ONE sample (scalar base + vector offset; the
//     generic kernels above spend a third of their VALU on 64-bit multiply-adds per tap) -- needs D*H*W*C*4 < 2^32;
//   * the second corner along an axis is the first + a conditional stride (no second multiply);
//   * the coefficient gradient first contracts the 4 channels of a lane with the gradient for each of the 8 corners
//     (32 FMAs), sums those 8 scalars over the voxel's lanes with DPP row shuffles, and only then forms the three
//     spatial derivatives -- instead of 3 x 4 difference-products per channel (144 ops);
//   * the tile walk of the gradient kernel is wave-uniform (scalar unit) except for a per-thread constant.
// Both are bound by the vector-memory path (8 x 64-byte gathers per output voxel through L1) once the VALU is lean.
typedef unsigned u32;

struct Tap32 {
  u32 o000, o001, o010, o011, o100, o101, o110, o111;          // byte offsets of the 8 corners (record start)
  float tx, ty, tz, mx, my, mz;
};

__device__ __forceinline__ void axis_tap(float g, int size, u32 stride, u32& o0, u32& d1, float& t, float& m) {
  float p;
  unnormalize_clip(g, size, p, m);
  if (!(p == p)) p = 0.f;                                        // NaN samples voxel 0 (ATen clip semantics)
  const float f = floorf(p);
  t = p - f;
  const int i0 = (int)f;
  o0 = (u32)i0 * stride;
  d1 = (i0 + 1 <= size - 1) ? stride : 0u;                       // clamped upper corner = same record
}

__device__ __forceinline__ Tap32 make_tap32(float gx, float gy, float gz, int W, int H, int D, u32 rec_bytes) {
  Tap32 t;
  u32 x0, dx, y0, dy, z0, dz;
  axis_tap(gx, W, rec_bytes, x0, dx, t.tx, t.mx);
  axis_tap(gy, H, rec_bytes * (u32)W, y0, dy, t.ty, t.my);
  axis_tap(gz, D, rec_bytes * (u32)W * (u32)H, z0, dz, t.tz, t.mz);
  const u32 b00 = z0 + y0, b01 = b00 + dy, b10 = b00 + dz, b11 = b01 + dz;
  t.o000 = b00 + x0; t.o001 = t.o000 + dx;
  t.o010 = b01 + x0; t.o011 = t.o010 + dx;
  t.o100 = b10 + x0; t.o101 = t.o100 + dx;
  t.o110 = b11 + x0; t.o111 = t.o110 + dx;
  return t;
}

__device__ __forceinline__ f32x4 ldrec(__amdgpu_buffer_rsrc_t rs, u32 off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}

// one thread per (voxel, 4-channel group); block = compact 2^lx x 2^ly x 2^lz tile; C % 4 == 0
template <int KIND>
__global__ void __launch_bounds__(256) resample_fwd_lean_kernel(
    const float* __restrict__ vol, long vol_bstride, const float* __restrict__ coef,
    float* __restrict__ out, int D, int H, int W, int C, int lpt, int lx, int ly, int lz, int nbz, Steps st) {
  const int lpv = C >> 2;
  const int n = blockIdx.z / nbz, bz = blockIdx.z - n * nbz;
  const int slot = threadIdx.x / lpt, q0 = threadIdx.x - slot * lpt;
  if (slot >= (1 << (lx + ly + lz))) return;
  const int x = (blockIdx.x << lx) + (slot & ((1 << lx) - 1));
  const int y = (blockIdx.y << ly) + ((slot >> lx) & ((1 << ly) - 1));
  const int z = (bz << lz) + (slot >> (lx + ly));
  if (x >= W || y >= H || z >= D) return;
  const u32 rec = (u32)C * 4u;
  const u32 sample_bytes = (u32)D * (u32)H * (u32)W * rec;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(vol + (long)n * vol_bstride), 0, sample_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)n * D * H * W * C), 0, sample_bytes, 0x00020000);
  const float* cf = coef + n * LF_MAP_COEFS;
  float gx, gy, gz, a, b, k;
  eval_grid<KIND>(cf, x, y, z, W, H, D, st, gx, gy, gz, a, b, k);
  const Tap32 t = make_tap32(gx, gy, gz, W, H, D, rec);
  const float wx1 = t.tx, wx0 = 1.f - t.tx, wy1 = t.ty, wy0 = 1.f - t.ty, wz1 = t.tz, wz0 = 1.f - t.tz;
  const float w000 = wx0 * wy0 * wz0, w001 = wx1 * wy0 * wz0, w010 = wx0 * wy1 * wz0, w011 = wx1 * wy1 * wz0;
  const float w100 = wx0 * wy0 * wz1, w101 = wx1 * wy0 * wz1, w110 = wx0 * wy1 * wz1, w111 = wx1 * wy1 * wz1;
  const u32 orow = (u32)((z * H + y) * W + x) * rec;
  for (int q = q0; q < lpv; q += lpt) {
    const u32 co = (u32)q * 16u;
    const f32x4 v000 = ldrec(rs, t.o000 + co), v001 = ldrec(rs, t.o001 + co), v010 = ldrec(rs, t.o010 + co), v011 = ldrec(rs, t.o011 + co);
    const f32x4 v100 = ldrec(rs, t.o100 + co), v101 = ldrec(rs, t.o101 + co), v110 = ldrec(rs, t.o110 + co), v111 = ldrec(rs, t.o111 + co);
    const f32x4 r = v000 * w000 + v001 * w001 + v010 * w010 + v011 * w011 + v100 * w100 + v101 * w101 + v110 * w110 + v111 * w111;
    // streamed output (nt): keep L2 for the gathered volume
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, r), ro, (int)(orow + co), 0, 2);
  }
}

// C == 16 specialisation of the lean gather (variant 3): fixed 4x4x4 tile, no integer division, no channel loop, one
// scalar-weight FMA per channel and corner (the generic form lets the compiler pack the FMAs in pairs, which costs
// a register move per weight to build the pairs).
// IO (round 5, training step under the bf16 storage policy): bit 0 -- the sampled volume, bit 1 -- the output are bf16
// channels-last records (32 B per voxel); the interpolation itself is the same fp32 arithmetic.
typedef __bf16 bf16x4r __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
template <bool B16>
__device__ __forceinline__ f32x4 ldrec_t(__amdgpu_buffer_rsrc_t rs, u32 off) {
  if constexpr (B16)
    return __builtin_convertvector(__builtin_bit_cast(bf16x4r, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0)), f32x4);
  else
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}

template <int KIND, int IO = 0>
__global__ void __launch_bounds__(256) resample_fwd_c16_kernel(
    const float* __restrict__ vol, long vol_bstride, const float* __restrict__ coef,
    float* __restrict__ out, int D, int H, int W, int nbz, Steps st) {
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
  constexpr u32 IREC = IN16 ? 32u : 64u, OREC = OUT16 ? 32u : 64u;
  const int n = blockIdx.z / nbz, bz = blockIdx.z - n * nbz;
  const int q = threadIdx.x & 3, vs = threadIdx.x >> 2;
  const int x = (blockIdx.x << 2) + (vs & 3), y = (blockIdx.y << 2) + ((vs >> 2) & 3), z = (bz << 2) + (vs >> 4);
  if (x >= W || y >= H || z >= D) return;
  const u32 nvox = (u32)D * (u32)H * (u32)W;
  // (vol_bstride counts ELEMENTS; the pointers are declared float*: byte arithmetic for the bf16 forms)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)vol + (long)n * vol_bstride * (IN16 ? 2 : 4)), 0, nvox * IREC, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)((char*)out + (long)n * nvox * OREC), 0, nvox * OREC, 0x00020000);
  const float* cf = coef + n * LF_MAP_COEFS;
  float gx, gy, gz, a, b, k;
  eval_grid<KIND>(cf, x, y, z, W, H, D, st, gx, gy, gz, a, b, k);
  const Tap32 t = make_tap32(gx, gy, gz, W, H, D, IREC);
  const u32 co = (u32)q * (IREC / 4u);
  const f32x4 v000 = ldrec_t<IN16>(rs, t.o000 + co), v001 = ldrec_t<IN16>(rs, t.o001 + co), v010 = ldrec_t<IN16>(rs, t.o010 + co), v011 = ldrec_t<IN16>(rs, t.o011 + co);
  const f32x4 v100 = ldrec_t<IN16>(rs, t.o100 + co), v101 = ldrec_t<IN16>(rs, t.o101 + co), v110 = ldrec_t<IN16>(rs, t.o110 + co), v111 = ldrec_t<IN16>(rs, t.o111 + co);
  const float wx1 = t.tx, wx0 = 1.f - t.tx, wy1 = t.ty, wy0 = 1.f - t.ty, wz1 = t.tz, wz0 = 1.f - t.tz;
  const float w000 = wx0 * wy0 * wz0, w001 = wx1 * wy0 * wz0, w010 = wx0 * wy1 * wz0, w011 = wx1 * wy1 * wz0;
  const float w100 = wx0 * wy0 * wz1, w101 = wx1 * wy0 * wz1, w110 = wx0 * wy1 * wz1, w111 = wx1 * wy1 * wz1;
  f32x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float acc = v000[e] * w000;
    // (inline asm keeps these as v_fmac_f32 with the weight as a plain operand)
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v001[e]), "v"(w001));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v010[e]), "v"(w010));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v011[e]), "v"(w011));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v100[e]), "v"(w100));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v101[e]), "v"(w101));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v110[e]), "v"(w110));
    asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v111[e]), "v"(w111));
    r[e] = acc;
  }
  if constexpr (OUT16)
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2_t, __builtin_convertvector(r, bf16x4r)), ro,
                                          (int)((u32)((z * H + y) * W + x) * 32u + (u32)q * 8u), 0, 2);
  else
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, r), ro, (int)((u32)((z * H + y) * W + x) * 64u + (u32)q * 16u), 0, 2);
}

// 16-channel gather with the per-voxel arithmetic done once per voxel (variant 5 of lf_set_tuning key 1).  The gather above
// is bound by VALU issue as much as by the L1 (PMC: VALU busy 93 % of its run time): map, clip and corner offsets are
// evaluated in all four lanes of a voxel.  Here a wave owns a 4x4x4 tile: phase A, lane = voxel, evaluates the 64 maps with
// one instruction stream and leaves (offset of corner 000, the three corner strides, the three fractions) in 2 KB of
// wave-private LDS; phase B, four passes of 16 voxels with lane = (voxel, channel quarter) as before, reads them back (one
// broadcast read per quad), forms the 8 weights and fetches the coalesced 64-byte records.  Same operations per voxel in the
// same order: bit-identical to resample_fwd_c16_kernel.
template <int KIND>
__global__ void __launch_bounds__(256) resample_fwd_c16_dedup_kernel(
    const float* __restrict__ vol, long vol_bstride, const float* __restrict__ coef,
    float* __restrict__ out, int D, int H, int W, int nbx, int nby, int nbz, Steps st) {
  __shared__ u32x4_t tapo[4][64];                                 // o000 | dead, dx, dy, dz (bytes; 0 if clamped)
  __shared__ f32x4 tapf[4][64];                                   // tx, ty, tz, -
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // tile index: 4 x-adjacent 4^3 tiles per workgroup (one per wave)
  const int n = blockIdx.z / nbz, bz = blockIdx.z - n * nbz;
  const int x0 = ((blockIdx.x << 2) + wave) << 2, y0 = blockIdx.y << 2, z0 = bz << 2;
  const u32 sample_bytes = (u32)D * (u32)H * (u32)W * 64u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(vol + (long)n * vol_bstride), 0, sample_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)n * D * H * W * 16), 0, sample_bytes, 0x00020000);
  const float* cf = coef + n * LF_MAP_COEFS;
  {
    const int x = x0 + (lane & 3), y = y0 + ((lane >> 2) & 3), z = z0 + (lane >> 4);
    const bool live = x < W && y < H && z < D;
    float gx, gy, gz, a, b, k;
    eval_grid<KIND>(cf, live ? x : 0, live ? y : 0, live ? z : 0, W, H, D, st, gx, gy, gz, a, b, k);
    const Tap32 t = make_tap32(gx, gy, gz, W, H, D, 64u);
    u32x4_t o;
    o[0] = live ? t.o000 : 0xffffffffu;
    o[1] = t.o001 - t.o000; o[2] = t.o010 - t.o000; o[3] = t.o100 - t.o000;
    tapo[wave][lane] = o;
    tapf[wave][lane] = (f32x4){t.tx, t.ty, t.tz, 0.f};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int q = lane & 3, vq = lane >> 2;
  const u32 co = (u32)q * 16u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int vi = 16 * i + vq;
    const u32x4_t o = tapo[wave][vi];
    const f32x4 f = tapf[wave][vi];
    if (o[0] == 0xffffffffu) continue;                            // voxel outside the volume (ragged tiles)
    const u32 b00 = o[0] + co, b01 = b00 + o[2], b10 = b00 + o[3], b11 = b01 + o[3];
    const f32x4 v000 = ldrec(rs, b00), v001 = ldrec(rs, b00 + o[1]), v010 = ldrec(rs, b01), v011 = ldrec(rs, b01 + o[1]);
    const f32x4 v100 = ldrec(rs, b10), v101 = ldrec(rs, b10 + o[1]), v110 = ldrec(rs, b11), v111 = ldrec(rs, b11 + o[1]);
    const float wx1 = f[0], wx0 = 1.f - f[0], wy1 = f[1], wy0 = 1.f - f[1], wz1 = f[2], wz0 = 1.f - f[2];
    const float w000 = wx0 * wy0 * wz0, w001 = wx1 * wy0 * wz0, w010 = wx0 * wy1 * wz0, w011 = wx1 * wy1 * wz0;
    const float w100 = wx0 * wy0 * wz1, w101 = wx1 * wy0 * wz1, w110 = wx0 * wy1 * wz1, w111 = wx1 * wy1 * wz1;
    f32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float acc = v000[e] * w000;
      asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v001[e]), "v"(w001));
      asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v010[e]), "v"(w010));
      asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v011[e]), "v"(w011));
      asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v100[e]), "v"(w100));
      asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v101[e]), "v"(w101));
      asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v110[e]), "v"(w110));
      asm("v_fmac_f32 %0, %1, %2" : "+v"(acc) : "v"(v111[e]), "v"(w111));
      r[e] = acc;
    }
    const int x = x0 + (vi & 3), y = y0 + ((vi >> 2) & 3), z = z0 + (vi >> 4);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, r), ro, (int)((u32)((z * H + y) * W + x) * 64u + co), 0, 2);
  }
}

// sum over the 4 lanes of a quad (all four receive it)
__device__ __forceinline__ float quad_sum4(float v) {
  const float a = v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));   // [1,0,3,2]
  return a + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a), 0x4E, 0xF, 0xF, true));            // [2,3,0,1]
}

// coefficient gradient, C == 16 (4 lanes per voxel, 64 voxels per block-iteration): block = 2^lg-voxel tile walked in
// 4x4x4 sub-tiles; thread (v = tid >> 2, q = tid & 3) keeps its position inside the sub-tile, the sub-tile index is
// wave-uniform
template <int MINW, int UNR>
__global__ void __launch_bounds__(256, MINW) resample_bwd_coef_c16_kernel(
    const float* __restrict__ gout, const float* __restrict__ vol, long vol_bstride,
    const float* __restrict__ coef, float* __restrict__ partial, int nblk, int vpb, BwdTile bt,
    int D, int H, int W, Steps st) {
  const unsigned fb = xcd_contiguous(blockIdx.x, gridDim.x);
  const int n = fb / nblk, blk = fb - n * nblk;
  const float* cf = coef + n * LF_MAP_COEFS;
  const u32 rec = 64u;
  const u32 sample_bytes = (u32)D * (u32)H * (u32)W * rec;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(vol + (long)n * vol_bstride), 0, sample_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(gout + (long)n * D * H * W * 16), 0, sample_bytes, 0x00020000);
  const int tx = blk % bt.ntx, ty = (blk / bt.ntx) % bt.nty, tz = blk / (bt.ntx * bt.nty);
  const int q = threadIdx.x & 3, vs = threadIdx.x >> 2;          // vs = position inside a 4x4x4 sub-tile
  const int px = vs & 3, py = (vs >> 2) & 3, pz = vs >> 4;
  const int sbx = bt.lx - 2, sby = bt.ly - 2;
  // the 18 sums (hx, hy, hz) x (1, a, b, k, ak, bk) are split over the voxel's four lanes -- after the quad sums every
  // lane holds the same (hx, hy, hz) -- so a lane carries 6 accumulators instead of 18: lane q owns basis functions
  // 2q and 2q+1 (lane 3 idles)
  float acc[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[i] = 0.f;
  const int nsub = vpb >> 6;
  const u32 co = (u32)q * 16u;
  // UNR sub-tiles are in flight per iteration: the 9 loads of a voxel have nothing to overlap with inside one
  // sub-tile (a block walks its tile serially), so with one sub-tile at a time the kernel is bound by memory latency,
  // not by bandwidth (measured: 0.64 ms at 4 waves/SIMD; the loads alone need ~0.3 ms of the L1 path)
  for (int sub0 = 0; sub0 < nsub; sub0 += UNR) {                 // wave-uniform
    Tap32 t[UNR];
    float a[UNR], b[UNR], k[UNR];
    f32x4 go[UNR], v[UNR][8];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int sub = sub0 + u;
      const int x = (tx << bt.lx) + ((sub & ((1 << sbx) - 1)) << 2) + px;
      const int y = (ty << bt.ly) + (((sub >> sbx) & ((1 << sby) - 1)) << 2) + py;
      const int z = (tz << bt.lz) + ((sub >> (sbx + sby)) << 2) + pz;
      const bool live = sub < nsub && x < W && y < H && z < D;
      float gx, gy, gz;
      eval_grid<LF_MAP_O2C>(cf, live ? x : 0, live ? y : 0, live ? z : 0, W, H, D, st, gx, gy, gz, a[u], b[u], k[u]);
      t[u] = make_tap32(gx, gy, gz, W, H, D, rec);
      // out-of-tile lanes read offset 0xffffffff: outside the descriptor's range -> zeros, no branch
      const u32 dead = live ? 0u : 0xffffffffu;
      go[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rg, (int)(((u32)((z * H + y) * W + x) * rec + co) | dead), 0, 2));      // streamed once (nt)
      v[u][0] = ldrec(rs, (t[u].o000 + co) | dead); v[u][1] = ldrec(rs, (t[u].o001 + co) | dead);
      v[u][2] = ldrec(rs, (t[u].o010 + co) | dead); v[u][3] = ldrec(rs, (t[u].o011 + co) | dead);
      v[u][4] = ldrec(rs, (t[u].o100 + co) | dead); v[u][5] = ldrec(rs, (t[u].o101 + co) | dead);
      v[u][6] = ldrec(rs, (t[u].o110 + co) | dead); v[u][7] = ldrec(rs, (t[u].o111 + co) | dead);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      float p[8];
#pragma unroll
      for (int c8 = 0; c8 < 8; ++c8)
        p[c8] = quad_sum4((go[u][0] * v[u][c8][0] + go[u][1] * v[u][c8][1]) + (go[u][2] * v[u][c8][2] + go[u][3] * v[u][c8][3]));
      const float wx1 = t[u].tx, wx0 = 1.f - wx1, wy1 = t[u].ty, wy0 = 1.f - wy1, wz1 = t[u].tz, wz0 = 1.f - wz1;
      // p index = z*4 + y*2 + x
      const float dxv = (p[1] - p[0]) * (wy0 * wz0) + (p[3] - p[2]) * (wy1 * wz0) + (p[5] - p[4]) * (wy0 * wz1) + (p[7] - p[6]) * (wy1 * wz1);
      const float dyv = (p[2] - p[0]) * (wx0 * wz0) + (p[3] - p[1]) * (wx1 * wz0) + (p[6] - p[4]) * (wx0 * wz1) + (p[7] - p[5]) * (wx1 * wz1);
      const float dzv = (p[4] - p[0]) * (wx0 * wy0) + (p[5] - p[1]) * (wx1 * wy0) + (p[6] - p[2]) * (wx0 * wy1) + (p[7] - p[3]) * (wx1 * wy1);
      const float hx = dxv * t[u].mx, hy = dyv * t[u].my, hz = dzv * t[u].mz;
      const float B0 = q == 0 ? 1.f : (q == 1 ? b[u] : (q == 2 ? a[u] * k[u] : 0.f));
      const float B1 = q == 0 ? a[u] : (q == 1 ? k[u] : (q == 2 ? b[u] * k[u] : 0.f));
      acc[0] += hx * B0; acc[1] += hy * B0; acc[2] += hz * B0;
      acc[3] += hx * B1; acc[4] += hy * B1; acc[5] += hz * B1;
    }
  }
  // workgroup reduction in fp64: a thread's fp32 sum runs over its 64 voxels only; from there on (64 lanes x 4 waves,
  // then the blocks in the finish kernel) nothing is rounded until the final conversion -- the 18 sums cancel to a
  // small fraction of their terms' magnitude, so summation rounding would otherwise show in the camera gradients
  __shared__ double red[4][18];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    double s = (double)acc[i];
#pragma unroll
    for (int o = 32; o >= 4; o >>= 1) s += __shfl_xor(s, o, 64);     // over the 16 lanes that share this q
    // lane q (0..2) of the first quad: basis 2q + i/3, component i%3 -> output index basis*3 + component
    if (lane < 3) red[wave][(2 * lane + i / 3) * 3 + (i % 3)] = s;
  }
  __syncthreads();
  if (threadIdx.x < 18) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    partial[((long)n * nblk + blk) * 18 + threadIdx.x] = (float)s;
  }
}

// coefficient gradient, C == 16, per-voxel arithmetic done ONCE per voxel (variant 6 of lf_set_tuning key 2).
// The kernel above spends most of its VALU issue on work that is identical in the four lanes sharing a voxel (map, taps,
// derivative algebra: ~3/4 of its ~200 instructions per lane and voxel), and VALU issue is what bounds it.
// Here a WAVE owns a 4x4x4 sub-tile and works on it in three phases that only meet through 3 KB of wave-private LDS:
//   A  lane = voxel (64 voxels per instruction): map, clip, corner offsets -> one 16-byte record per voxel in LDS
//      (fractions, clip masks and lattice coordinates stay in the lane's registers for phase C);
//   B  four passes of 16 voxels, lane = (voxel, channel quarter) as before: the gathers stay coalesced 64-byte records
//      through L1 (a lane-per-voxel gather would quadruple the L1 look-ups), contraction with the gradient, quad sums;
//      the 8 per-corner scalars of a voxel go back to LDS;
//   C  lane = voxel again: spatial derivatives, clip masks, the 18 basis sums.
// No workgroup barrier; waves walk their own sub-tiles.  Same value as the kernel above, different summation order.
template <int PIF, int MINW, bool GOPF = false>
__global__ void __launch_bounds__(256, MINW) resample_bwd_coef_c16_dedup_kernel(
    const float* __restrict__ gout, const float* __restrict__ vol, long vol_bstride,
    const float* __restrict__ coef, float* __restrict__ partial, int nblk, int vpb, BwdTile bt,
    int D, int H, int W, Steps st) {
  __shared__ u32x4_t tapbuf[4][64];                             // per wave: o000 | dead, x / y / z corner strides (bytes, 0 if clamped)
  __shared__ float pbuf[4][64 * 8];                             // per wave: p[corner] of every voxel
  __shared__ double red[4][18];
  const unsigned fb = xcd_contiguous(blockIdx.x, gridDim.x);
  const int n = fb / nblk, blk = fb - n * nblk;
  const float* cf = coef + n * LF_MAP_COEFS;
  const u32 rec = 64u;
  const u32 sample_bytes = (u32)D * (u32)H * (u32)W * rec;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(vol + (long)n * vol_bstride), 0, sample_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(gout + (long)n * D * H * W * 16), 0, sample_bytes, 0x00020000);
  const int tx = blk % bt.ntx, ty = (blk / bt.ntx) % bt.nty, tz = blk / (bt.ntx * bt.nty);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = lane & 3, py = (lane >> 2) & 3, pz = lane >> 4;  // phase A / C: lane = voxel of the sub-tile
  const int q = lane & 3, vq = lane >> 2;                         // phase B: lane = (voxel 16 i + vq, quarter q)
  const int sbx = bt.lx - 2, sby = bt.ly - 2;
  const int nsub = vpb >> 6;
  const u32 co = (u32)q * 16u;
  float acc[18];
#pragma unroll
  for (int i = 0; i < 18; ++i) acc[i] = 0.f;
  u32x4_t* tb = tapbuf[wave];
  float* pb = pbuf[wave];
  // GOPF: the gradient records (streamed from HBM: the longest latency of an iteration; the gathered volume sits in the
  // Infinity Cache) are requested one sub-tile ahead
  f32x4 gnext[4];
  auto load_go = [&](int sub_, f32x4 (&dst)[4]) {
    const int bx0 = (tx << bt.lx) + ((sub_ & ((1 << sbx) - 1)) << 2);
    const int by0 = (ty << bt.ly) + (((sub_ >> sbx) & ((1 << sby) - 1)) << 2);
    const int bz0 = (tz << bt.lz) + ((sub_ >> (sbx + sby)) << 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vi = 16 * i + vq;
      const int vx = bx0 + (vi & 3), vy = by0 + ((vi >> 2) & 3), vz = bz0 + (vi >> 4);
      const u32 dead = (sub_ < nsub && vx < W && vy < H && vz < D) ? 0u : 0xffffffffu;
      dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
          rg, (int)(((u32)((vz * H + vy) * W + vx) * rec + co) | dead), 0, 2));
    }
  };
  if (GOPF) load_go(wave, gnext);
  for (int sub = wave; sub < nsub; sub += 4) {                    // wave-uniform
    const int x0 = (tx << bt.lx) + ((sub & ((1 << sbx) - 1)) << 2);
    const int y0 = (ty << bt.ly) + (((sub >> sbx) & ((1 << sby) - 1)) << 2);
    const int z0 = (tz << bt.lz) + ((sub >> (sbx + sby)) << 2);
    f32x4 gcur[4];
    if (GOPF) {
#pragma unroll
      for (int i = 0; i < 4; ++i) gcur[i] = gnext[i];
      load_go(sub + 4, gnext);
    }
    // ---- A ----
    const int x = x0 + px, y = y0 + py, z = z0 + pz;
    const bool live = x < W && y < H && z < D;
    float gx, gy, gz, a, b, k;
    eval_grid<LF_MAP_O2C>(cf, live ? x : 0, live ? y : 0, live ? z : 0, W, H, D, st, gx, gy, gz, a, b, k);
    u32 ox, dx, oy, dy, oz, dz;
    float tx_, ty_, tz_, mx, my, mz;
    axis_tap(gx, W, rec, ox, dx, tx_, mx);
    axis_tap(gy, H, rec * (u32)W, oy, dy, ty_, my);
    axis_tap(gz, D, rec * (u32)W * (u32)H, oz, dz, tz_, mz);
    u32x4_t tr;
    tr[0] = live ? (oz + oy + ox) : 0xffffffffu;                  // dead voxels: every offset out of range -> zeros
    tr[1] = live ? dx : 0u; tr[2] = live ? dy : 0u; tr[3] = live ? dz : 0u;
    tb[lane] = tr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- B ---- (PIF passes in flight; a real loop, so that no more than PIF x 9 loads are live)
#pragma unroll 1
    for (int i0 = 0; i0 < 4; i0 += (GOPF ? 4 : PIF)) {
      if (GOPF) {
        // passes unrolled (static indices into gcur), PIF gathers in flight
#pragma unroll
        for (int j0 = 0; j0 < 4; j0 += PIF) {
          f32x4 v[PIF][8];
#pragma unroll
          for (int u = 0; u < PIF; ++u) {
            const u32x4_t t = tb[16 * (j0 + u) + vq];
            const u32 dead = t[0] == 0xffffffffu ? 0xffffffffu : 0u;
            const u32 b00 = (t[0] + co) | dead, b01 = b00 + t[2], b10 = b00 + t[3], b11 = b01 + t[3];
            v[u][0] = ldrec(rs, b00); v[u][1] = ldrec(rs, b00 + t[1]);
            v[u][2] = ldrec(rs, b01); v[u][3] = ldrec(rs, b01 + t[1]);
            v[u][4] = ldrec(rs, b10); v[u][5] = ldrec(rs, b10 + t[1]);
            v[u][6] = ldrec(rs, b11); v[u][7] = ldrec(rs, b11 + t[1]);
          }
#pragma unroll
          for (int u = 0; u < PIF; ++u) {
            const f32x4 go = gcur[j0 + u];
            float p[8];
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8)
              p[c8] = quad_sum4((go[0] * v[u][c8][0] + go[1] * v[u][c8][1]) + (go[2] * v[u][c8][2] + go[3] * v[u][c8][3]));
            const float e0 = q == 0 ? p[0] : (q == 1 ? p[2] : (q == 2 ? p[4] : p[6]));
            const float e1 = q == 0 ? p[1] : (q == 1 ? p[3] : (q == 2 ? p[5] : p[7]));
            *(float2*)(pb + (16 * (j0 + u) + vq) * 8 + 2 * q) = make_float2(e0, e1);
          }
          __builtin_amdgcn_sched_barrier(0);                      // keep at most PIF x 8 gathers live
        }
        continue;
      }
      f32x4 go[PIF], v[PIF][8];
#pragma unroll
      for (int u = 0; u < PIF; ++u) {
        const int vi = 16 * (i0 + u) + vq;                        // voxel of the sub-tile: same numbering as phase A's lane
        const u32x4_t t = tb[vi];
        const int vx = x0 + (vi & 3), vy = y0 + ((vi >> 2) & 3), vz = z0 + (vi >> 4);
        const u32 dead = t[0] == 0xffffffffu ? 0xffffffffu : 0u;
        go[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
            rg, (int)(((u32)((vz * H + vy) * W + vx) * rec + co) | dead), 0, 2));        // streamed once (nt)
        const u32 b00 = (t[0] + co) | dead, b01 = b00 + t[2], b10 = b00 + t[3], b11 = b01 + t[3];
        v[u][0] = ldrec(rs, b00); v[u][1] = ldrec(rs, b00 + t[1]);
        v[u][2] = ldrec(rs, b01); v[u][3] = ldrec(rs, b01 + t[1]);
        v[u][4] = ldrec(rs, b10); v[u][5] = ldrec(rs, b10 + t[1]);
        v[u][6] = ldrec(rs, b11); v[u][7] = ldrec(rs, b11 + t[1]);
      }
#pragma unroll
      for (int u = 0; u < PIF; ++u) {
        float p[8];
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8)
          p[c8] = quad_sum4((go[u][0] * v[u][c8][0] + go[u][1] * v[u][c8][1]) + (go[u][2] * v[u][c8][2] + go[u][3] * v[u][c8][3]));
        // every lane of a quad holds the voxel's 8 sums; lane q stores corners 2q, 2q+1 (one contiguous 32 bytes per voxel)
        const float e0 = q == 0 ? p[0] : (q == 1 ? p[2] : (q == 2 ? p[4] : p[6]));
        const float e1 = q == 0 ? p[1] : (q == 1 ? p[3] : (q == 2 ? p[5] : p[7]));
        *(float2*)(pb + (16 * (i0 + u) + vq) * 8 + 2 * q) = make_float2(e0, e1);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- C ----
    {
      const f32x4 pa = *(const f32x4*)(pb + lane * 8), pc = *(const f32x4*)(pb + lane * 8 + 4);
      const float wx1 = tx_, wx0 = 1.f - wx1, wy1 = ty_, wy0 = 1.f - wy1, wz1 = tz_, wz0 = 1.f - wz1;
      // corner index = z*4 + y*2 + x: pa = corners 0..3 (z0), pc = corners 4..7 (z1)
      const float dxv = (pa[1] - pa[0]) * (wy0 * wz0) + (pa[3] - pa[2]) * (wy1 * wz0) + (pc[1] - pc[0]) * (wy0 * wz1) + (pc[3] - pc[2]) * (wy1 * wz1);
      const float dyv = (pa[2] - pa[0]) * (wx0 * wz0) + (pa[3] - pa[1]) * (wx1 * wz0) + (pc[2] - pc[0]) * (wx0 * wz1) + (pc[3] - pc[1]) * (wx1 * wz1);
      const float dzv = (pc[0] - pa[0]) * (wx0 * wy0) + (pc[1] - pa[1]) * (wx1 * wy0) + (pc[2] - pa[2]) * (wx0 * wy1) + (pc[3] - pa[3]) * (wx1 * wy1);
      const float hx = live ? dxv * mx : 0.f, hy = live ? dyv * my : 0.f, hz = live ? dzv * mz : 0.f;
      const float ak = a * k, bk = b * k;
      acc[0] += hx;       acc[1] += hy;       acc[2] += hz;
      acc[3] += hx * a;   acc[4] += hy * a;   acc[5] += hz * a;
      acc[6] += hx * b;   acc[7] += hy * b;   acc[8] += hz * b;
      acc[9] += hx * k;   acc[10] += hy * k;  acc[11] += hz * k;
      acc[12] += hx * ak; acc[13] += hy * ak; acc[14] += hz * ak;
      acc[15] += hx * bk; acc[16] += hy * bk; acc[17] += hz * bk;
    }
    asm volatile("" ::: "memory");                               // the next sub-tile's phase A overwrites tb after phase B's reads (program order)
  }
#pragma unroll
  for (int i = 0; i < 18; ++i) {
    double s = (double)acc[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) red[wave][i] = s;
  }
  __syncthreads();
  if (threadIdx.x < 18) {
    const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    partial[((long)n * nblk + blk) * 18 + threadIdx.x] = (float)s;
  }
}

#include "resample_staged.inc"

#ifdef STAGED_TPW_OVERRIDE
constexpr int STAGED_TPW = STAGED_TPW_OVERRIDE;
#else
constexpr int STAGED_TPW = 16;   // tiles per workgroup of the staged coefficient gradient
#endif

int g_bwd_coef_variant = 10;   // lean coefficient gradient: 1 = one sub-tile in flight (6 waves/SIMD), 2 = two (4 waves/SIMD; r02 default),
                              // 3 = as 2 with a 128-register cap, 4 / 5 = one in flight capped at 6 / 8 waves/SIMD
int g_resample_variant = 3;   // 1 = generic kernels, 2 = lean kernels, 3 = lean + 16-channel gather, 4 = LDS-staged footprint (16 channels;
                              // other shapes as 3) (lf_set_tuning)

// Sample evaluation for the deterministic splats with floating-point contraction OFF: every operation is rounded on its own,
// so the kernels that must agree with each other -- the bounding-box pass and the tile pass of the tiled form (a corner voxel
// outside its block's box would be dropped), and the tiled and the atomic form -- get the same corner indices, fractions and
// weights no matter how the surrounding code is scheduled (with contraction on, the backend fuses multiply-adds per call site).
struct SplatTap { int x0, y0, z0, x1, y1, z1; float w[8]; };     // w index = z*4 + y*2 + x

template <int KIND>
__device__ __forceinline__ SplatTap splat_eval(const float* __restrict__ cf, int x, int y, int z, int W, int H, int D, Steps st) {
#pragma clang fp contract(off)
  const float a = (x < W / 2) ? st.w * (float)x : 1.f - st.w * (float)(W - 1 - x);
  const float b = (y < H / 2) ? st.h * (float)y : 1.f - st.h * (float)(H - 1 - y);
  const float k = (z < D / 2) ? st.d * (float)z : 1.f - st.d * (float)(D - 1 - z);
  float g[3];
  if (KIND == LF_MAP_O2C) {
    const float ak = a * k, bk = b * k;
#pragma unroll
    for (int c = 0; c < 3; ++c) g[c] = ((((cf[c] + cf[3 + c] * a) + cf[6 + c] * b) + cf[9 + c] * k) + cf[12 + c] * ak) + cf[15 + c] * bk;
  } else {
    const float lx = 2.f * a - 1.f, ly = 2.f * b - 1.f, lz = 2.f * k - 1.f;
    const float n0 = ((cf[0] * lx + cf[1] * ly) + cf[2] * lz) + cf[3];
    const float n1 = ((cf[4] * lx + cf[5] * ly) + cf[6] * lz) + cf[7];
    const float n2 = ((cf[8] * lx + cf[9] * ly) + cf[10] * lz) + cf[11];
    const float dn = ((cf[12] * lx + cf[13] * ly) + cf[14] * lz) + cf[15];
    g[0] = n0 / dn;
    g[1] = n1 / dn;
    g[2] = n2;
  }
  const int size[3] = {W, H, D};
  int i0[3], i1[3];
  float t[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float p = ((g[c] + 1.f) * (float)size[c] - 1.f) * 0.5f;         // grid_sampler_unnormalize (align_corners=False)
    p = fminf(fmaxf(p, 0.f), (float)(size[c] - 1));                 // border clip
    if (!(p == p)) p = 0.f;                                         // NaN samples voxel 0 (ATen clip semantics)
    const float f = floorf(p);
    t[c] = p - f;
    i0[c] = (int)f;
    i1[c] = min(i0[c] + 1, size[c] - 1);
  }
  SplatTap s;
  s.x0 = i0[0]; s.y0 = i0[1]; s.z0 = i0[2]; s.x1 = i1[0]; s.y1 = i1[1]; s.z1 = i1[2];
  const float wx[2] = {1.f - t[0], t[0]}, wy[2] = {1.f - t[1], t[1]}, wz[2] = {1.f - t[2], t[2]};
#pragma unroll
  for (int i = 0; i < 8; ++i) s.w[i] = (wx[i & 1] * wy[(i >> 1) & 1]) * wz[i >> 2];
  return s;
}

// The same evaluation shared by the four lanes of a quad that work on ONE sample (the binned tile pass: a lane quad per list entry,
// round 6): lane q < 3 evaluates axis q -- its numerator, for the projective axes the denominator and the division, the un-normalise /
// clip / floor chain -- lane 3 repeats axis 2; three quad broadcasts (v_mov_b32 dpp quad_perm) per value hand every lane all three
// axes.  Every operation an axis sees is the one splat_eval performs for it, in the same order, contraction off: the same bits, for
// about 55 % of the instructions (the evaluation was a quarter of the tile pass, profiles/r06_splat_ab.txt).
template <int CTRL>
__device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ float quad_bcast_f(float v) { return __int_as_float(quad_bcast_i<CTRL>(__float_as_int(v))); }

template <int KIND>
__device__ __forceinline__ SplatTap splat_eval_quad(const float* __restrict__ cf, int x, int y, int z, int W, int H, int D, Steps st, int q) {
#pragma clang fp contract(off)
  const float a = (x < W / 2) ? st.w * (float)x : 1.f - st.w * (float)(W - 1 - x);
  const float b = (y < H / 2) ? st.h * (float)y : 1.f - st.h * (float)(H - 1 - y);
  const float k = (z < D / 2) ? st.d * (float)z : 1.f - st.d * (float)(D - 1 - z);
  const int c = q < 3 ? q : 2;                                     // this lane's axis
  float g;
  if (KIND == LF_MAP_O2C) {
    const float ak = a * k, bk = b * k;
    g = ((((cf[c] + cf[3 + c] * a) + cf[6 + c] * b) + cf[9 + c] * k) + cf[12 + c] * ak) + cf[15 + c] * bk;
  } else {
    const float lx = 2.f * a - 1.f, ly = 2.f * b - 1.f, lz = 2.f * k - 1.f;
    const float num = ((cf[4 * c] * lx + cf[4 * c + 1] * ly) + cf[4 * c + 2] * lz) + cf[4 * c + 3];
    const float dn = ((cf[12] * lx + cf[13] * ly) + cf[14] * lz) + cf[15];
    g = c < 2 ? num / dn : num;
  }
  const int size = c == 0 ? W : (c == 1 ? H : D);
  float p = ((g + 1.f) * (float)size - 1.f) * 0.5f;               // grid_sampler_unnormalize (align_corners=False)
  p = fminf(fmaxf(p, 0.f), (float)(size - 1));                    // border clip
  if (!(p == p)) p = 0.f;                                         // NaN samples voxel 0 (ATen clip semantics)
  const float f = floorf(p);
  const float tm = p - f;
  const int i0m = (int)f;
  const int i1m = min(i0m + 1, size - 1);
  SplatTap s;
  s.x0 = quad_bcast_i<0x00>(i0m); s.y0 = quad_bcast_i<0x55>(i0m); s.z0 = quad_bcast_i<0xaa>(i0m);
  s.x1 = quad_bcast_i<0x00>(i1m); s.y1 = quad_bcast_i<0x55>(i1m); s.z1 = quad_bcast_i<0xaa>(i1m);
  const float t0 = quad_bcast_f<0x00>(tm), t1 = quad_bcast_f<0x55>(tm), t2 = quad_bcast_f<0xaa>(tm);
  const float wx[2] = {1.f - t0, t0}, wy[2] = {1.f - t1, t1}, wz[2] = {1.f - t2, t2};
#pragma unroll
  for (int i = 0; i < 8; ++i) s.w[i] = (wx[i & 1] * wy[(i >> 1) & 1]) * wz[i >> 2];
  return s;
}

// ---- deterministic splat: the same scatter, accumulated in 64-bit fixed point ----------------------------------
// Float atomics make the result depend on the order in which the hardware retires them.  Integer addition is
// associative, so accumulating round(contribution * 2^K) with 64-bit integer atomics gives bit-identical results
// for every execution order; K is chosen from max|gout| so that 2^24 contributions (every output voxel of every
// sample landing on ONE source voxel -- border clamping can do that) cannot overflow: the quantum is 2^-38 of the
// largest gradient, finer than fp32's own resolution of any sum it could be added to.
__global__ void __launch_bounds__(256) absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f && m < 3.0e38f) atomicMax(out, __float_as_uint(m));   // max is order-independent
}

// (n a multiple of 16 and x 16-byte aligned: channels-last 16-channel records; a lane reads 8 values per load)
__global__ void __launch_bounds__(256) absmax_bf16_kernel(const __bf16* __restrict__ x, long n, unsigned* __restrict__ out) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n / 8; i += (long)gridDim.x * 256) {
    const u32x4 r = ((const u32x4*)x)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                  // |bf16| as fp32: the 16 bits shifted into the high half, sign cleared
      m = fmaxf(m, __uint_as_float((r[k] << 16) & 0x7fffffffu));
      m = fmaxf(m, __uint_as_float(r[k] & 0x7fff0000u));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f && m < 3.0e38f) atomicMax(out, __float_as_uint(m));
}

__device__ __forceinline__ float fixed_scale(const unsigned* amax) {
  const float am = __uint_as_float(*amax);
  if (!(am > 0.f)) return 1.f;
  int ex;
  frexpf(am, &ex);                                               // am = m * 2^ex, m in [0.5, 1)
  // |contribution| * scale < 2^38; the exponent is capped so that tiny gradients (max|g| < 2^-88) keep a FINITE scale
  // (2^126: they then simply use fewer of the 64 bits) instead of inf * 0 = NaN
  return ldexpf(1.f, min(38 - ex, 126));
}

// round-to-nearest-even of v (|v| < 2^39) as a 64-bit integer, in 8 VALU instructions instead of the ~20 of the generic
// float -> int64 conversion (half of the tiled kernel's arithmetic): r = rint(v) is an integer-valued float; its value
// splits exactly into hi * 2^24 + lo with lo in [0, 2^24), both exact in fp32 and in range of the 32-bit converts.
#ifndef LF_FIXED_MAGIC
#define LF_FIXED_MAGIC 1
#endif
__device__ __forceinline__ unsigned long long fixed_round(float v) {
#if LF_FIXED_MAGIC
  // round 6: the same integer in 3 instructions.  v is exact in fp64; adding 1.5 * 2^52 leaves a double in [2^52, 2^53) whose
  // unit in the last place is 1, so the fp64 add itself rounds v to the nearest integer (ties to even, the mode of rintf) and
  // the mantissa field then holds 2^51 + that integer.  Subtracting the bit pattern of the constant (its low word is zero: one
  // 32-bit add on the high word) leaves the integer in two's complement.
  const double d = (double)v + 6755399441055744.0;
  return (unsigned long long)__double_as_longlong(d) - 0x4338000000000000ull;
#endif
  const float r = __builtin_rintf(v);
  const float hi = __builtin_floorf(r * 5.9604644775390625e-08f);          // 2^-24
  const float lo = __builtin_fmaf(hi, -16777216.f, r);
  const int hi_i = (int)hi;
  const unsigned lo_u = (unsigned)lo;
  return ((unsigned long long)(unsigned)(hi_i >> 8) << 32) | (unsigned)(((unsigned)hi_i << 24) | lo_u);
}

template <int KIND>
__global__ void __launch_bounds__(256) resample_bwd_vol_fixed_kernel(
    const float* __restrict__ gout, const float* __restrict__ coef, unsigned long long* __restrict__ acc,
    long acc_bstride, const unsigned* __restrict__ amax, int N, int D, int H, int W, int C, Steps st) {
  const long per_sample = (long)D * H * W * C;
  const int n = blockIdx.y;
  const float* cf = coef + (long)n * LF_MAP_COEFS;
  const float scale = fixed_scale(amax);
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < per_sample; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C);
    long v = idx / C;
    const int x = (int)(v % W); v /= W;
    const int y = (int)(v % H);
    const int z = (int)(v / H);
    const SplatTap t = splat_eval<KIND>(cf, x, y, z, W, H, D, st);
    const float go = gout[(long)n * per_sample + idx] * scale;   // exact: power-of-two scale
    unsigned long long* base = acc + (long)n * acc_bstride + c;
    const long sW = C, sH = (long)W * C, sD = (long)H * W * C;
#define SPLAT(Z, Y, X, WI) atomicAdd(base + (Z) * sD + (Y) * sH + (X) * sW, (unsigned long long)__float2ll_rn(go * t.w[WI]))
    SPLAT(t.z0, t.y0, t.x0, 0); SPLAT(t.z0, t.y0, t.x1, 1);
    SPLAT(t.z0, t.y1, t.x0, 2); SPLAT(t.z0, t.y1, t.x1, 3);
    SPLAT(t.z1, t.y0, t.x0, 4); SPLAT(t.z1, t.y0, t.x1, 5);
    SPLAT(t.z1, t.y1, t.x0, 6); SPLAT(t.z1, t.y1, t.x1, 7);
#undef SPLAT
  }
}

__global__ void __launch_bounds__(256) fixed_to_float_kernel(const long long* __restrict__ acc, const unsigned* __restrict__ amax,
                                                            float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = (float)((double)acc[i] * (1.0 / (double)fixed_scale(amax)));   // (the scale is a power of two: its reciprocal is exact)
}

// ---- the same sums without global atomics (C == 16): every SOURCE tile is owned by one workgroup ---------------------
// The fixed-point scatter above is bound by the L2's atomic units (2.1e9 64-bit atomics per 8 x 128^3 x 16 launch: 16-19 ms,
// a quarter of a training step).  Integer addition being associative, the same totals can be formed in any grouping:
//   pass 1  one wave per 4x4x4 block of OUTPUT voxels: bounding box of the (clamped) corner voxels its samples touch;
//   pass 2  one workgroup per 4x8x8 tile of SOURCE voxels: 64-bit accumulators for the tile in LDS (32 KB); it culls the
//           output blocks in two levels (boxes of 16^3 super-blocks, then the 64 block boxes of those that overlap; a
//           ballot per 64 boxes, every wave redundantly: no exchange, no barriers), re-evaluates the
//           samples of the blocks that touch it and adds the contributions that land inside the tile with LDS atomics;
//           border clamping needs no special case (the boxes are boxes of clamped indices); with one volume shared by all
//           samples (vol_n == 1) the workgroup walks all samples, so their contributions meet in the same accumulators;
//           finally the tile is converted and written with plain stores.
// Same quantisation, same integer totals, same conversion => bit-identical to the atomic kernel.
constexpr int STZ = 4, STY = 8, STX = 8;                          // source tile owned by a workgroup: 256 voxels
#ifndef SACC
#define SACC 17                                                   // 64-bit accumulators per voxel record (16 + padding)
#endif

template <int KIND>
__global__ void __launch_bounds__(256) splat_bbox_kernel(const float* __restrict__ coef, uint3* __restrict__ bbox, int nblk, int nbx,
                                                         int nby, int D, int H, int W, Steps st) {
  const int lane = threadIdx.x & 63;
  const int blk = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y;
  if (blk >= nblk) return;                                        // (wave-uniform)
  const int bx = blk % nbx, by = (blk / nbx) % nby, bz = blk / (nbx * nby);
  const int x = bx * 4 + (lane & 3), y = by * 4 + ((lane >> 2) & 3), z = bz * 4 + (lane >> 4);
  const bool live = x < W && y < H && z < D;
  const SplatTap t = splat_eval<KIND>(coef + (long)n * LF_MAP_COEFS, live ? x : 0, live ? y : 0, live ? z : 0, W, H, D, st);
  int lo[3] = {live ? t.x0 : 0x7fff, live ? t.y0 : 0x7fff, live ? t.z0 : 0x7fff};
  int hi[3] = {live ? t.x1 : -1, live ? t.y1 : -1, live ? t.z1 : -1};
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lo[c] = min(lo[c], __shfl_xor(lo[c], o, 64));
      hi[c] = max(hi[c], __shfl_xor(hi[c], o, 64));
    }
  if (lane == 0)                                                  // (an empty block: lo = 0x7fff > hi = 0xffff as signed 16 bit = -1)
    bbox[(long)n * nblk + blk] = make_uint3((unsigned)lo[0] | ((unsigned)(hi[0] & 0xffff) << 16), (unsigned)lo[1] | ((unsigned)(hi[1] & 0xffff) << 16),
                                            (unsigned)lo[2] | ((unsigned)(hi[2] & 0xffff) << 16));
}

// union of the boxes of the 4x4x4 blocks of a super-block (16^3 output voxels): one wave per super-block
__global__ void __launch_bounds__(256) splat_bbox2_kernel(const uint3* __restrict__ bbox, uint3* __restrict__ sbox, int nblk, int nsb,
                                                          int nbx, int nby, int nbz, int nsx, int nsy) {
  const int lane = threadIdx.x & 63;
  const int sb = blockIdx.x * 4 + (threadIdx.x >> 6), n = blockIdx.y;
  if (sb >= nsb) return;                                          // (wave-uniform)
  const int bx = (sb % nsx) * 4 + (lane & 3), by = ((sb / nsx) % nsy) * 4 + ((lane >> 2) & 3), bz = (sb / (nsx * nsy)) * 4 + (lane >> 4);
  int lo[3] = {0x7fff, 0x7fff, 0x7fff}, hi[3] = {-1, -1, -1};
  if (bx < nbx && by < nby && bz < nbz) {
    const uint3 r = bbox[(long)n * nblk + ((long)bz * nby + by) * nbx + bx];
    lo[0] = (short)(r.x & 0xffff); hi[0] = (short)(r.x >> 16);
    lo[1] = (short)(r.y & 0xffff); hi[1] = (short)(r.y >> 16);
    lo[2] = (short)(r.z & 0xffff); hi[2] = (short)(r.z >> 16);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lo[c] = min(lo[c], __shfl_xor(lo[c], o, 64));
      hi[c] = max(hi[c], __shfl_xor(hi[c], o, 64));
    }
  if (lane == 0)
    sbox[(long)n * nsb + sb] = make_uint3((unsigned)(lo[0] & 0xffff) | ((unsigned)(hi[0] & 0xffff) << 16),
                                          (unsigned)(lo[1] & 0xffff) | ((unsigned)(hi[1] & 0xffff) << 16),
                                          (unsigned)(lo[2] & 0xffff) | ((unsigned)(hi[2] & 0xffff) << 16));
}

template <int KIND, int IO = 0>                                   // IO: bit 0 -- gout, bit 1 -- gvol stored as bf16 records
__global__ void __launch_bounds__(256) splat_tile_kernel(const float* __restrict__ gout, const float* __restrict__ coef,
                                                         const uint3* __restrict__ bbox, const uint3* __restrict__ sbox,
                                                         const unsigned* __restrict__ amax, float* __restrict__ gvol, int vol_n, int N,
                                                         int nblk, int nsb, int nbx, int nby, int nbz, int nsx, int nsy, int ntx, int nty,
                                                         int D, int H, int W, Steps st) {
  __shared__ unsigned long long acc[STZ * STY * STX * SACC];       // 34 KB: records padded to 17 (an 128-byte stride puts the
                                                                   // 16 voxels of an atomic instruction on two sets of banks)
  const int tid = threadIdx.x, lane = tid & 63;
  const int tile = blockIdx.x;
  const int tx0 = (tile % ntx) * STX, ty0 = ((tile / ntx) % nty) * STY, tz0 = (tile / (ntx * nty)) * STZ;
  for (int i = tid; i < STZ * STY * STX * SACC; i += 256) acc[i] = 0ull;
  const float scale = fixed_scale(amax);
  const long nvox = (long)D * H * W;
  const int q = tid & 3, v = tid >> 2;                             // lane quad = one output voxel of the block, 4 channels each
  const int px = v & 3, py = (v >> 2) & 3, pz = v >> 4;
  const int n_first = vol_n == 1 ? 0 : blockIdx.y, n_last = vol_n == 1 ? N : blockIdx.y + 1;
  auto overlaps = [&](const uint3 r) {
    const int x0 = (short)(r.x & 0xffff), x1 = (short)(r.x >> 16), y0 = (short)(r.y & 0xffff), y1 = (short)(r.y >> 16),
              z0 = (short)(r.z & 0xffff), z1 = (short)(r.z >> 16);
    return x0 < tx0 + STX && x1 >= tx0 && y0 < ty0 + STY && y1 >= ty0 && z0 < tz0 + STZ && z1 >= tz0;
  };
  __syncthreads();
  // Two-level culling, done redundantly by every wave (same data -> same masks -> same walk; no LDS exchange, no barriers:
  // the LDS atomics commute): 64 super-block boxes per test, then the 64 block boxes of a super-block that overlaps the tile.
  for (int n = n_first; n < n_last; ++n) {
    const float* cf = coef + (long)n * LF_MAP_COEFS;
    const float* gs = (const float*)((const char*)gout + (long)n * nvox * ((IO & 1) ? 32 : 64));
    const uint3* bb = bbox + (long)n * nblk;
    const uint3* sbb = sbox + (long)n * nsb;
    for (int sbase = 0; sbase < nsb; sbase += 64) {
      unsigned long long sm = __ballot(sbase + lane < nsb && overlaps(sbb[min(sbase + lane, nsb - 1)]));
      while (sm) {
        const int sbit = __builtin_ctzll(sm);
        sm &= sm - 1;
        const int sb = sbase + sbit;
        const int cbx = (sb % nsx) * 4 + (lane & 3), cby = ((sb / nsx) % nsy) * 4 + ((lane >> 2) & 3), cbz = (sb / (nsx * nsy)) * 4 + (lane >> 4);
        const bool cok = cbx < nbx && cby < nby && cbz < nbz;
        const long cid = ((long)cbz * nby + cby) * nbx + cbx;
        unsigned long long cm = __ballot(cok && overlaps(bb[cok ? cid : 0]));
        while (cm) {
          const int cbit = __builtin_ctzll(cm);
          cm &= cm - 1;
          const int bx = (sb % nsx) * 4 + (cbit & 3), by = ((sb / nsx) % nsy) * 4 + ((cbit >> 2) & 3), bz = (sb / (nsx * nsy)) * 4 + (cbit >> 4);
          const int x = bx * 4 + px, y = by * 4 + py, z = bz * 4 + pz;
          if (x < W && y < H && z < D) {
            const SplatTap t = splat_eval<KIND>(cf, x, y, z, W, H, D, st);
            f32x4 g4;
            if constexpr ((IO & 1) != 0)
              g4 = __builtin_convertvector(*(const bf16x4r*)((const char*)gs + (((long)z * H + y) * W + x) * 32 + q * 8), f32x4);
            else
              g4 = *(const f32x4*)(gs + (((long)z * H + y) * W + x) * 16 + q * 4);
#define SPLAT_T(Z, Y, X, WI) do { \
              const int lz_ = (Z) - tz0, ly_ = (Y) - ty0, lx_ = (X) - tx0; \
              if ((unsigned)lz_ < (unsigned)STZ && (unsigned)ly_ < (unsigned)STY && (unsigned)lx_ < (unsigned)STX) { \
                unsigned long long* d_ = acc + ((lz_ * STY + ly_) * STX + lx_) * SACC + q * 4; \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) atomicAdd(d_ + e, fixed_round((g4[e] * scale) * t.w[WI])); \
              } } while (0)
            SPLAT_T(t.z0, t.y0, t.x0, 0); SPLAT_T(t.z0, t.y0, t.x1, 1);
            SPLAT_T(t.z0, t.y1, t.x0, 2); SPLAT_T(t.z0, t.y1, t.x1, 3);
            SPLAT_T(t.z1, t.y0, t.x0, 4); SPLAT_T(t.z1, t.y0, t.x1, 5);
            SPLAT_T(t.z1, t.y1, t.x0, 6); SPLAT_T(t.z1, t.y1, t.x1, 7);
#undef SPLAT_T
          }
        }
      }
    }
  }
  __syncthreads();
  // the tile: one thread per voxel, 16 channels = 4 float4 stores
  const int lx = tid % STX, ly = (tid / STX) % STY, lz = tid / (STX * STY);
  const int x = tx0 + lx, y = ty0 + ly, z = tz0 + lz;
  if (x < W && y < H && z < D) {
    constexpr int OREC = (IO & 2) ? 32 : 64;
    char* dst = (char*)gvol + ((vol_n == 1 ? 0 : (long)blockIdx.y * nvox) + (((long)z * H + y) * W + x)) * OREC;
    const double inv = 1.0 / (double)scale;                       // exact: the scale is a power of two (2^-90 .. 2^126)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (float)((double)(long long)acc[tid * SACC + c4 * 4 + e] * inv);
      if constexpr ((IO & 2) != 0) *(bf16x4r*)(dst + c4 * 8) = __builtin_convertvector(o, bf16x4r);
      else *(f32x4*)(dst + c4 * 16) = o;
    }
  }
}

// ---- binned form of the same sums (round 5: the camera -> object splat of the training step, a volume per sample, and the
// object -> camera one, all samples into one volume).  The tile form above finds the output voxels that touch a source tile by culling boxes: at 128^3 a workgroup
// walks 512 super-block boxes and then re-evaluates 40-75 blocks of 64 samples of which a tenth lands in its tile (12.4 ms for
// 32 views, the largest kernel of the step).  Here the output voxels are BINNED by the source tiles their corners touch first
// (count, scan, fill: 1.6 list entries per voxel on average, 8 at most), and a tile's workgroup evaluates exactly its list:
//   bin<FILL = false>  per output voxel: the <= 8 distinct tiles of its corner voxels, one wave-aggregated atomic add per tile
//                      and wave on the tile's counter;
//   scan               exclusive prefix of a sample's tile counters;
//   bin<FILL = true>   the same walk, now storing the voxel (z << 20 | y << 10 | x) at offset[tile] + slot;
//   binned tile        64-bit LDS accumulators as above, one LANE per list entry (all 16 channels), conversion and store as above.
// The order of a list depends on the atomics; the integer sums do not: bit-identical to the other two forms.
template <int KIND, bool FILL>
__global__ void __launch_bounds__(256) splat_bin_kernel(const float* __restrict__ coef, unsigned* __restrict__ cnt,
                                                        const unsigned* __restrict__ off, unsigned* __restrict__ list, int n0, long nvox,
                                                        long cap, int ntiles, int ntx, int nty, int D, int H, int W, Steps st) {
  const int lane = threadIdx.x & 63;
  const int nl = blockIdx.y;                                       // sample within the chunk
  // a wave takes a 4x4x4 block of output voxels: its samples land in one or two tiles per axis (a row of 64 voxels would
  // cross four to eight), so the aggregation loop below runs once or twice per corner combination
  const int nbx = (W + 3) >> 2, nby = (H + 3) >> 2, nbz = (D + 3) >> 2;
  const long blk = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool wave_live = blk < (long)nbx * nby * nbz;
  const long bq = wave_live ? blk : 0;
  const int x = (int)(bq % nbx) * 4 + (lane & 3), y = (int)((bq / nbx) % nby) * 4 + ((lane >> 2) & 3), z = (int)(bq / ((long)nbx * nby)) * 4 + (lane >> 4);
  const bool live = wave_live && x < W && y < H && z < D;
  const SplatTap t = splat_eval<KIND>(coef + (long)(n0 + nl) * LF_MAP_COEFS, min(x, W - 1), min(y, H - 1), min(z, D - 1), W, H, D, st);
  const int tx[2] = {t.x0 / STX, t.x1 / STX}, ty[2] = {t.y0 / STY, t.y1 / STY}, tz[2] = {t.z0 / STZ, t.z1 / STZ};
  unsigned* c = cnt + (long)nl * ntiles;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int kx = k & 1, ky = (k >> 1) & 1, kz = k >> 2;
    // a combination is a NEW tile iff every axis it takes the upper corner on really changes tile there
    const bool act = live && (!kx || tx[1] != tx[0]) && (!ky || ty[1] != ty[0]) && (!kz || tz[1] != tz[0]);
    const int tile = (tz[kz] * nty + ty[ky]) * ntx + tx[kx];
    unsigned long long todo = __ballot(act);
    while (todo) {                                                 // (wave-uniform loop: one atomic per distinct tile and wave)
      const int leader = __builtin_ctzll(todo);
      const int lt = __shfl(tile, leader, 64);
      const unsigned long long same = __ballot(act && tile == lt);
      unsigned base = 0;
      if (lane == leader) base = atomicAdd(c + lt, (unsigned)__builtin_popcountll(same));
      if (FILL) {
        base = __shfl(base, leader, 64);
        if (act && tile == lt) {
          const unsigned slot = base + (unsigned)__builtin_popcountll(same & ((1ull << lane) - 1ull));
          list[(long)nl * cap + off[(long)nl * ntiles + lt] + slot] = ((unsigned)z << 20) | ((unsigned)y << 10) | (unsigned)x;
        }
      }
      todo &= ~same;
    }
  }
}

// exclusive prefix sums of each sample's tile counters (one workgroup per sample)
__global__ void __launch_bounds__(256) splat_scan_kernel(const unsigned* __restrict__ cnt, unsigned* __restrict__ off, int ntiles) {
  __shared__ unsigned part[256];
  const unsigned* c = cnt + (long)blockIdx.x * ntiles;
  unsigned* o = off + (long)blockIdx.x * ntiles;
  const int per = (ntiles + 255) / 256, b = threadIdx.x * per, e = min(b + per, ntiles);
  unsigned sum = 0;
  for (int i = b; i < e; ++i) sum += c[i];
  part[threadIdx.x] = sum;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int i = 0; i < 256; ++i) { const unsigned p = part[i]; part[i] = run; run += p; }
  }
  __syncthreads();
  unsigned run = part[threadIdx.x];
  for (int i = b; i < e; ++i) { o[i] = run; run += c[i]; }
}

// threads per source tile: 512 = eight waves share the tile's accumulators, 32 waves per CU at four resident workgroups (LDS-bound
// residency): 2.39 -> 2.34 ms (training geometry), 2.06 -> 1.93 ms (one shared volume), 3.25 -> 3.14 ms (wide) against 256
#ifndef SBT_THREADS
#define SBT_THREADS 512
#endif
template <int KIND, int IO>
__global__ void __launch_bounds__(SBT_THREADS) splat_binned_tile_kernel(const float* __restrict__ gout, const float* __restrict__ coef,
                                                                const unsigned* __restrict__ cnt, const unsigned* __restrict__ off,
                                                                const unsigned* __restrict__ list, const unsigned* __restrict__ amax,
                                                                float* __restrict__ gvol, int n0, int m, int shared, long nvox, long cap,
                                                                int ntiles, int ntx, int nty, int D, int H, int W, Steps st) {
  __shared__ unsigned long long acc[STZ * STY * STX * SACC];
  const int tid = threadIdx.x;
  const int tile = blockIdx.x;
  // a volume per sample: blockIdx.y is the sample; one volume shared by the m samples: the workgroup walks their m lists
  const int nl_first = shared ? 0 : (int)blockIdx.y, nl_last = shared ? m : (int)blockIdx.y + 1;
  const long n_out = shared ? 0 : n0 + (long)blockIdx.y;
  const int tx0 = (tile % ntx) * STX, ty0 = ((tile / ntx) % nty) * STY, tz0 = (tile / (ntx * nty)) * STZ;
  unsigned total = 0;
  for (int nl = nl_first; nl < nl_last; ++nl) total |= cnt[(long)nl * ntiles + tile];
  if (total == 0) {
    if (tid >= STZ * STY * STX) return;                            // nothing lands here (more than half of the camera volume's tiles
    const int lx = tid % STX, ly = (tid / STX) % STY, lz = tid / (STX * STY);   // when the object fills part of it): zeros, no LDS pass
    const int x = tx0 + lx, y = ty0 + ly, z = tz0 + lz;
    if (x < W && y < H && z < D) {
      constexpr int OREC = (IO & 2) ? 32 : 64;
      f32x4* dst = (f32x4*)((char*)gvol + (n_out * nvox + (((long)z * H + y) * W + x)) * OREC);
#pragma unroll
      for (int k = 0; k < OREC / 16; ++k) dst[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  for (int i = tid; i < STZ * STY * STX * SACC; i += SBT_THREADS) acc[i] = 0ull;
  const float scale = fixed_scale(amax);
  __syncthreads();
  // a lane quad per list entry, four channels each (as the tile form: the lanes of one atomic instruction then spread over 16
  // records x 4 slots; one lane per entry measured 2.7x slower -- clamped samples pile up on border voxels and 64 lanes on one
  // address serialise); corners of weight zero (the far corner on an axis a sample was clamped on) add nothing and are skipped
  const int q = tid & 3;
  for (int nl = nl_first; nl < nl_last; ++nl) {
    const float* cf = coef + (long)(n0 + nl) * LF_MAP_COEFS;
    const char* gs = (const char*)gout + (long)(n0 + nl) * nvox * ((IO & 1) ? 32 : 64);
    const unsigned count = cnt[(long)nl * ntiles + tile];
    const unsigned* mine = list + (long)nl * cap + off[(long)nl * ntiles + tile];
    // two list entries ahead, one gradient record ahead (round 6): the chain list word -> record address -> record is two HBM / L2
    // latencies long and a workgroup holds only four waves; with the loads of the next entries in flight behind the arithmetic of
    // this one the pass is no longer latency-bound (tools/splat_ab.py)
    typedef typename std::conditional<(IO & 1) != 0, bf16x4r, f32x4>::type graw_t;
    auto g_of = [&](unsigned pk_) -> graw_t {
      const long v_ = ((long)(pk_ >> 20) * H + (long)((pk_ >> 10) & 1023u)) * W + (long)(pk_ & 1023u);
      return *(const graw_t*)(gs + v_ * ((IO & 1) ? 32 : 64) + q * ((IO & 1) ? 8 : 16));
    };
    const unsigned i0 = tid >> 2;
    unsigned pk1 = i0 < count ? mine[i0] : 0u, pk2 = i0 + SBT_THREADS / 4 < count ? mine[i0 + SBT_THREADS / 4] : 0u;
    graw_t gnext = g_of(pk1);
    constexpr unsigned EPI = SBT_THREADS / 4;                      // entries per iteration of the workgroup
    for (unsigned i = i0; i < count; i += EPI) {
      const unsigned pk = pk1;                                      // z << 20 | y << 10 | x
      const graw_t graw = gnext;
      pk1 = pk2;
      if (i + EPI < count) gnext = g_of(pk1);
      pk2 = i + 2 * EPI < count ? mine[i + 2 * EPI] : 0u;
      const int x = (int)(pk & 1023u), y = (int)((pk >> 10) & 1023u), z = (int)(pk >> 20);
      const SplatTap t = splat_eval_quad<KIND>(cf, x, y, z, W, H, D, st, q);   // (the quad's lanes are all live or all past the list's end)
      f32x4 g4;
      if constexpr ((IO & 1) != 0) g4 = __builtin_convertvector(graw, f32x4);
      else g4 = graw;
      g4 = g4 * scale;
#define SPLAT_B(Z, Y, X, WI) do { \
        const int lz_ = (Z) - tz0, ly_ = (Y) - ty0, lx_ = (X) - tx0; \
        if (t.w[WI] != 0.f && (unsigned)lz_ < (unsigned)STZ && (unsigned)ly_ < (unsigned)STY && (unsigned)lx_ < (unsigned)STX) { \
          unsigned long long* d_ = acc + ((lz_ * STY + ly_) * STX + lx_) * SACC + q * 4; \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) atomicAdd(d_ + e, fixed_round(g4[e] * t.w[WI])); \
        } } while (0)
      SPLAT_B(t.z0, t.y0, t.x0, 0); SPLAT_B(t.z0, t.y0, t.x1, 1);
      SPLAT_B(t.z0, t.y1, t.x0, 2); SPLAT_B(t.z0, t.y1, t.x1, 3);
      SPLAT_B(t.z1, t.y0, t.x0, 4); SPLAT_B(t.z1, t.y0, t.x1, 5);
      SPLAT_B(t.z1, t.y1, t.x0, 6); SPLAT_B(t.z1, t.y1, t.x1, 7);
#undef SPLAT_B
    }
  }
  __syncthreads();
  const int lx = tid % STX, ly = (tid / STX) % STY, lz = tid / (STX * STY);
  const int x = tx0 + lx, y = ty0 + ly, z = tz0 + lz;
  if (tid < STZ * STY * STX && x < W && y < H && z < D) {
    constexpr int OREC = (IO & 2) ? 32 : 64;
    char* dst = (char*)gvol + (n_out * nvox + (((long)z * H + y) * W + x)) * OREC;
    const double inv = 1.0 / (double)scale;                       // exact: the scale is a power of two (2^-90 .. 2^126)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (float)((double)(long long)acc[tid * SACC + c4 * 4 + e] * inv);
      if constexpr ((IO & 2) != 0) *(bf16x4r*)(dst + c4 * 8) = __builtin_convertvector(o, bf16x4r);
      else *(f32x4*)(dst + c4 * 16) = o;
    }
  }
}

// ---- round 6 A/B (NOT the default: lf_set_tuning(4, 4)): the binned tile pass with ONE LANE PER CHANNEL (16 lanes per entry) ------
// tools/ub/lds_atomic2.hip: 64-bit LDS atomics retire at 8-9 lane-atomics per clock and CU whatever the shape of the access -- as
// long as the lanes of one instruction do not meet on an address.  The quad-per-entry form above puts 16 list entries into one
// instruction; neighbouring entries come from one 4x4x4 block of output voxels and land on the same few source voxels, clamped
// samples pile up on border records: on ONE record it falls to 2.0 per clock.  With 16 lanes per entry an instruction covers
// 4 entries x 128 contiguous bytes, which the LDS serves at 8.0 per clock even when all four are the SAME record.  To keep the
// arithmetic per entry from growing 4x with the lanes, a wave works in two phases per 64 entries: (A) lane-per-entry: list word,
// the sample's gradient record, splat_eval, in-tile test -> a table in LDS (8 weights, 8 record numbers or 0xffff, the record);
// (B) 16 lanes per entry read their entry's row (broadcast reads) and issue the 8 adds.  Same quantisation, same integer totals,
// same conversion => bit-identical to the other forms.
// MEASURED (profiles/r06_splat_ab.txt, 8 x 128^3 x 16, training geometry): 2.82 ms against 2.50 ms of the quad form before its
// loads were pipelined (2.36 after).  Ablations of THIS kernel: without the atomics -0.30 ms, without splat_eval -0.42 ms, without
// the conversion -0.07 ms of 1.9 ms: the atomics were never the bound -- the dependent loads (list word -> record) and the
// per-entry arithmetic at four waves per workgroup are; which is what the pipelined loads in the quad form address.
template <int KIND, int IO>
__global__ void __launch_bounds__(256) splat_binned_tile16_kernel(const float* __restrict__ gout, const float* __restrict__ coef,
                                                                  const unsigned* __restrict__ cnt, const unsigned* __restrict__ off,
                                                                  const unsigned* __restrict__ list, const unsigned* __restrict__ amax,
                                                                  float* __restrict__ gvol, int n0, int m, int shared, long nvox, long cap,
                                                                  int ntiles, int ntx, int nty, int D, int H, int W, Steps st) {
  constexpr bool IN16 = (IO & 1) != 0, OUT16 = (IO & 2) != 0;
  constexpr int NREC = STZ * STY * STX, GREC = IN16 ? 32 : 64;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  __shared__ unsigned long long acc[NREC * 16];                    // 32 KB: record stride 128 B (the shape is conflict-free as it is)
  __shared__ __attribute__((aligned(16))) float tw[4][64][8];      // per wave and entry: the 8 corner weights
  __shared__ __attribute__((aligned(16))) unsigned short tr[4][64][8];   // the 8 records (0xffff: outside the tile or weight zero)
  __shared__ __attribute__((aligned(16))) unsigned char tg[4][64][GREC]; // the sample's 16-channel gradient record as stored
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, grp = lane >> 4, ch = lane & 15;
  const int tile = blockIdx.x;
  const int nl_first = shared ? 0 : (int)blockIdx.y, nl_last = shared ? m : (int)blockIdx.y + 1;
  const long n_out = shared ? 0 : n0 + (long)blockIdx.y;
  const int tx0 = (tile % ntx) * STX, ty0 = ((tile / ntx) % nty) * STY, tz0 = (tile / (ntx * nty)) * STZ;
  constexpr int OREC = OUT16 ? 32 : 64;
  unsigned total = 0;
  for (int nl = nl_first; nl < nl_last; ++nl) total |= cnt[(long)nl * ntiles + tile];
  if (total == 0) {
    const int lx = tid % STX, ly = (tid / STX) % STY, lz = tid / (STX * STY);
    const int x = tx0 + lx, y = ty0 + ly, z = tz0 + lz;
    if (x < W && y < H && z < D) {
      f32x4* dst = (f32x4*)((char*)gvol + (n_out * nvox + (((long)z * H + y) * W + x)) * OREC);
#pragma unroll
      for (int k = 0; k < OREC / 16; ++k) dst[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    return;
  }
  for (int i = tid; i < NREC * 16; i += 256) acc[i] = 0ull;
  const float scale = fixed_scale(amax);
  __syncthreads();
  for (int nl = nl_first; nl < nl_last; ++nl) {
    const float* cf = coef + (long)(n0 + nl) * LF_MAP_COEFS;
    const char* gs = (const char*)gout + (long)(n0 + nl) * nvox * GREC;
    const unsigned count = cnt[(long)nl * ntiles + tile];
    const unsigned* mine = list + (long)nl * cap + off[(long)nl * ntiles + tile];
    for (unsigned b0 = (unsigned)wv * 64u; b0 < count; b0 += 256u) {
      // ---- phase A: lane = list entry
      if (b0 + lane < count) {
        const unsigned pk = mine[b0 + lane];                        // z << 20 | y << 10 | x
        const int x = (int)(pk & 1023u), y = (int)((pk >> 10) & 1023u), z = (int)(pk >> 20);
        const u32x4* src = (const u32x4*)(gs + (((long)z * H + y) * W + x) * GREC);
        u32x4 rec[GREC / 16];
#pragma unroll
        for (int k = 0; k < GREC / 16; ++k) rec[k] = src[k];
        const SplatTap t = splat_eval<KIND>(cf, x, y, z, W, H, D, st);
        const int lz[2] = {t.z0 - tz0, t.z1 - tz0}, ly[2] = {t.y0 - ty0, t.y1 - ty0}, lx[2] = {t.x0 - tx0, t.x1 - tx0};
        unsigned r16[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int cz = lz[c >> 2], cy = ly[(c >> 1) & 1], cx = lx[c & 1];
          const bool in = t.w[c] != 0.f && (unsigned)cz < (unsigned)STZ && (unsigned)cy < (unsigned)STY && (unsigned)cx < (unsigned)STX;
          r16[c] = in ? (unsigned)((cz * STY + cy) * STX + cx) : 0xffffu;
        }
        *(f32x4*)&tw[wv][lane][0] = (f32x4){t.w[0], t.w[1], t.w[2], t.w[3]};
        *(f32x4*)&tw[wv][lane][4] = (f32x4){t.w[4], t.w[5], t.w[6], t.w[7]};
        *(u32x4*)&tr[wv][lane][0] = (u32x4){r16[0] | (r16[1] << 16), r16[2] | (r16[3] << 16), r16[4] | (r16[5] << 16), r16[6] | (r16[7] << 16)};
#pragma unroll
        for (int k = 0; k < GREC / 16; ++k) *(u32x4*)&tg[wv][lane][k * 16] = rec[k];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- phase B: 16 lanes = the channels of one entry, four entries per instruction
      const int nb = (int)min(64u, count - b0);
#pragma unroll 4
      for (int j = 0; j < 16; ++j) {
        const int e = j * 4 + grp;
        if (e < nb) {
          const f32x4 w0 = *(const f32x4*)&tw[wv][e][0], w1 = *(const f32x4*)&tw[wv][e][4];
          const u32x4 r4 = *(const u32x4*)&tr[wv][e][0];
          float g;
          if constexpr (IN16) g = __uint_as_float((unsigned)(*(const unsigned short*)&tg[wv][e][ch * 2]) << 16);
          else g = *(const float*)&tg[wv][e][ch * 4];
          g = g * scale;
          const float w8[8] = {w0[0], w0[1], w0[2], w0[3], w1[0], w1[1], w1[2], w1[3]};
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const unsigned r = (r4[c >> 1] >> (16 * (c & 1))) & 0xffffu;
            if (r != 0xffffu) atomicAdd(acc + r * 16 + ch, fixed_round(g * w8[c]));
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();                             // (the table is rewritten by the next batch)
    }
  }
  __syncthreads();
  // conversion and store, 16 lanes per record (a record's 128 bytes in one access; four x-neighbours per instruction)
  const double inv = 1.0 / (double)scale;                       // exact: the scale is a power of two (2^-90 .. 2^126)
#pragma unroll 4
  for (int j = 0; j < 16; ++j) {
    const int r = wv * 64 + j * 4 + grp;
    const int lx = r % STX, ly = (r / STX) % STY, lz = r / (STX * STY);
    const int x = tx0 + lx, y = ty0 + ly, z = tz0 + lz;
    if (x < W && y < H && z < D) {
      char* dst = (char*)gvol + (n_out * nvox + (((long)z * H + y) * W + x)) * OREC;
      const float o = (float)((double)(long long)acc[r * 16 + ch] * inv);
      if constexpr (OUT16) ((__bf16*)dst)[ch] = __builtin_convertvector((f32x4){o, o, o, o}, bf16x4r)[0];
      else ((float*)dst)[ch] = o;
    }
  }
}

// samples per pass of the binned form: the lists are sized for the worst case (8 entries per voxel), 512 MB at most
int g_splat_chunk_cap = 0;                                        // lf_set_tuning key 6: samples per pass at most (0 = by memory only)
inline int splat_bin_chunk(int N, long nvox) {
  long cv = (512L << 20) / (nvox * 32);
  if (cv < 1) cv = 1;
  if (g_splat_chunk_cap > 0 && cv > g_splat_chunk_cap) cv = g_splat_chunk_cap;
  return (int)(cv < N ? cv : N);
}

int g_splat_variant = 2;      // deterministic splat (lf_set_tuning key 4): 1 = global 64-bit atomics, 2 = source tiles in LDS (C == 16;
                              // lf_resample3d_bwd_vol_det_io: binned lists, a lane quad per list entry), 3 = as 2 without the binned form,
                              // 4 = as 2 with 16 lanes per list entry (round-6 A/B: slower)

bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

Steps make_steps(int D, int H, int W) {
  Steps st;
  st.w = W > 1 ? 1.0f / (float)(W - 1) : 0.f;
  st.h = H > 1 ? 1.0f / (float)(H - 1) : 0.f;
  st.d = D > 1 ? 1.0f / (float)(D - 1) : 0.f;
  return st;
}

}  // namespace

extern "C" int lf_resample3d_fwd(const float* vol, int vol_n, const float* coef, int kind, float* out,
                                 int N, int D, int H, int W, int C, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return LF_EINVAL;
  if (vol_n != 1 && vol_n != N) return LF_EINVAL;
  if (kind != LF_MAP_O2C && kind != LF_MAP_C2O) return LF_EINVAL;
  const long bstride = vol_n == 1 ? 0 : (long)D * H * W * C;
  if ((long)D * H * W >= 0x7fffffffL) return LF_EINVAL;
  const bool vec = (C % 4 == 0) && lf_aligned16(vol) && lf_aligned16(out);
  const int lpv = vec ? C / 4 : C;
  const int lpt = lpv < 256 ? lpv : 256;
  int lg = 0;                                            // log2(voxels per block), rounded down
  while ((2 << lg) * lpt <= 256) ++lg;
  int tlx, tly, tlz;                                     // log2 tile extents
  if (lg >= 6) { tlx = (lg + 2) / 3; tly = (lg + 1) / 3; tlz = lg / 3; }      // 6:(2,2,2) 7:(3,2,2) 8:(3,3,2)
  else         { tlx = lg < 2 ? lg : 2; tly = lg - tlx < 2 ? lg - tlx : 2; tlz = lg - tlx - tly; }   // 5:(2,2,1) 4:(2,2,0) ...
  const int nbz = (D + (1 << tlz) - 1) >> tlz;
  if ((long)nbz * N > 65535 || ((H + (1 << tly) - 1) >> tly) > 65535) return LF_EINVAL;
  dim3 grid((unsigned)((W + (1 << tlx) - 1) >> tlx), (unsigned)((H + (1 << tly) - 1) >> tly), (unsigned)(nbz * N)), block(256);
  const Steps st = make_steps(D, H, W);
  hipStream_t s = (hipStream_t)stream;
  if (g_resample_variant == 4 && vec && C == 16 && (long)D * H * W * 64 < 0xffffffffL && W <= 65535) {
    const int ntx = (W + LT_X - 1) / LT_X, nty = (H + LT_Y - 1) / LT_Y, ntz = (D + LT_Z - 1) / LT_Z;
    const long nwg = (long)ntx * nty * ntz * N;
    if (nwg > 0x7fffffffL) return LF_EINVAL;
    if (kind == LF_MAP_O2C)
      hipLaunchKernelGGL((resample_fwd_staged_kernel<LF_MAP_O2C>), dim3((unsigned)nwg), block, 0, s, vol, bstride, coef, out, D, H, W, ntx, nty, ntz, st);
    else
      hipLaunchKernelGGL((resample_fwd_staged_kernel<LF_MAP_C2O>), dim3((unsigned)nwg), block, 0, s, vol, bstride, coef, out, D, H, W, ntx, nty, ntz, st);
    return lf_launch_status();
  }
  if (g_resample_variant == 5 && vec && C == 16 && (long)D * H * W * 64 < 0xffffffffL) {
    const int nbz4 = (D + 3) >> 2, nbx16 = (W + 15) >> 4, nby4 = (H + 3) >> 2;
    if ((long)nbz4 * N > 65535 || nby4 > 65535) return LF_EINVAL;
    dim3 g5((unsigned)nbx16, (unsigned)nby4, (unsigned)(nbz4 * N));
    if (kind == LF_MAP_O2C)
      hipLaunchKernelGGL((resample_fwd_c16_dedup_kernel<LF_MAP_O2C>), g5, block, 0, s, vol, bstride, coef, out, D, H, W, nbx16, nby4, nbz4, st);
    else
      hipLaunchKernelGGL((resample_fwd_c16_dedup_kernel<LF_MAP_C2O>), g5, block, 0, s, vol, bstride, coef, out, D, H, W, nbx16, nby4, nbz4, st);
    return lf_launch_status();
  }
  if (g_resample_variant >= 3 && vec && C == 16 && (long)D * H * W * 64 < 0xffffffffL) {
    const int nbz4 = (D + 3) >> 2;
    if ((long)nbz4 * N > 65535 || ((H + 3) >> 2) > 65535) return LF_EINVAL;
    dim3 g4((unsigned)((W + 3) >> 2), (unsigned)((H + 3) >> 2), (unsigned)(nbz4 * N));
    if (kind == LF_MAP_O2C)
      hipLaunchKernelGGL((resample_fwd_c16_kernel<LF_MAP_O2C>), g4, block, 0, s, vol, bstride, coef, out, D, H, W, nbz4, st);
    else
      hipLaunchKernelGGL((resample_fwd_c16_kernel<LF_MAP_C2O>), g4, block, 0, s, vol, bstride, coef, out, D, H, W, nbz4, st);
    return lf_launch_status();
  }
  if (g_resample_variant >= 2 && vec && (long)D * H * W * C * 4 < 0xffffffffL) {
    if (kind == LF_MAP_O2C)
      hipLaunchKernelGGL((resample_fwd_lean_kernel<LF_MAP_O2C>), grid, block, 0, s, vol, bstride, coef, out, D, H, W, C, lpt, tlx, tly, tlz, nbz, st);
    else
      hipLaunchKernelGGL((resample_fwd_lean_kernel<LF_MAP_C2O>), grid, block, 0, s, vol, bstride, coef, out, D, H, W, C, lpt, tlx, tly, tlz, nbz, st);
    return lf_launch_status();
  }
#define LAUNCH(K, V) hipLaunchKernelGGL((resample_fwd_kernel<K, V>), grid, block, 0, s, vol, bstride, coef, out, N, D, H, W, C, lpt, tlx, tly, tlz, nbz, st)
  if (kind == LF_MAP_O2C) { if (vec) LAUNCH(LF_MAP_O2C, 4); else LAUNCH(LF_MAP_O2C, 1); }
  else                    { if (vec) LAUNCH(LF_MAP_C2O, 4); else LAUNCH(LF_MAP_C2O, 1); }
#undef LAUNCH
  return lf_launch_status();
}

static long staged_bwd_blocks(int D, int H, int W) {
  const long nt = (long)((W + LT_X - 1) / LT_X) * ((H + LT_Y - 1) / LT_Y) * ((D + LT_Z - 1) / LT_Z);
  return (nt + STAGED_TPW - 1) / STAGED_TPW;
}

extern "C" size_t lf_resample3d_bwd_coef_scratch_bytes(int N, int D, int H, int W) {
  const long nvox = (long)D * H * W;
  const int vpb = bwd_vox_per_block(nvox, N);
  const BwdTile bt = bwd_tile(vpb, D, H, W);
  const long nblk = (long)bt.ntx * bt.nty * bt.ntz;
  const long nblk_staged = staged_bwd_blocks(D, H, W);             // whichever form runs (lf_set_tuning) must fit
  return (size_t)N * (nblk > nblk_staged ? nblk : nblk_staged) * 18 * sizeof(float);
}

extern "C" int lf_resample3d_bwd_coef(const float* gout, const float* vol, int vol_n, const float* coef,
                                      float* gcoef, void* scratch, size_t scratch_bytes,
                                      int N, int D, int H, int W, int C, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return LF_EINVAL;
  if (vol_n != 1 && vol_n != N) return LF_EINVAL;
  if (scratch_bytes < lf_resample3d_bwd_coef_scratch_bytes(N, D, H, W)) return LF_ENOSPC;
  if ((long)D * H * W >= 0x7fffffffL) return LF_EINVAL;
  const long nvox = (long)D * H * W;
  const int vpb = bwd_vox_per_block(nvox, N);
  const BwdTile bt = bwd_tile(vpb, D, H, W);
  const long nblk_l = (long)bt.ntx * bt.nty * bt.ntz;
  if (nblk_l * N > 0x7fffffffL) return LF_EINVAL;
  const int nblk = (int)nblk_l;
  const long bstride = vol_n == 1 ? 0 : nvox * C;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(nblk_l * N)), block(256);
  float* partial = (float*)scratch;
  // lanes per voxel: next power of two covering the channel groups (idle lanes contribute 0)
  const bool vec = (C % 4 == 0) && lf_aligned16(gout) && lf_aligned16(vol);
  const int groups = vec ? C / 4 : C;
  int lpv = 1;
  while (lpv < groups) lpv <<= 1;
  if (lpv > 64) return LF_EINVAL;                       // C > 256 (vec) / C > 64 (scalar)
  if (g_resample_variant == 4 && vec && C == 16 && nvox * 64 < 0xffffffffL && W <= 65535) {
    const int ntx = (W + LT_X - 1) / LT_X, nty = (H + LT_Y - 1) / LT_Y, ntz = (D + LT_Z - 1) / LT_Z;
    const long nb = staged_bwd_blocks(D, H, W);
    if (nb * N > 0x7fffffffL) return LF_EINVAL;
    hipLaunchKernelGGL((resample_bwd_coef_staged_kernel<STAGED_TPW>), dim3((unsigned)(nb * N)), block, 0, s, gout, vol, bstride, coef,
                       partial, (int)nb, D, H, W, ntx, nty, ntz, make_steps(D, H, W));
    int st4 = lf_launch_status();
    if (st4) return st4;
    hipLaunchKernelGGL(resample_bwd_coef_reduce, dim3(N), dim3(256), 0, s, partial, (int)nb, gcoef);
    return lf_launch_status();
  }
  if (g_resample_variant >= 2 && vec && C == 16 && vpb >= 64 && nvox * 64 < 0xffffffffL) {
    const Steps stp = make_steps(D, H, W);
    if (g_bwd_coef_variant == 1)
      hipLaunchKernelGGL((resample_bwd_coef_c16_kernel<1, 1>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 2)
      hipLaunchKernelGGL((resample_bwd_coef_c16_kernel<1, 2>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 3)
      hipLaunchKernelGGL((resample_bwd_coef_c16_kernel<4, 2>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 6 && vpb >= 256)
      hipLaunchKernelGGL((resample_bwd_coef_c16_dedup_kernel<2, 1>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 7 && vpb >= 256)
      hipLaunchKernelGGL((resample_bwd_coef_c16_dedup_kernel<1, 1>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 8 && vpb >= 256)
      hipLaunchKernelGGL((resample_bwd_coef_c16_dedup_kernel<2, 4>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 9 && vpb >= 256)
      hipLaunchKernelGGL((resample_bwd_coef_c16_dedup_kernel<1, 5>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 10 && vpb >= 256)
      hipLaunchKernelGGL((resample_bwd_coef_c16_dedup_kernel<2, 1, true>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 11 && vpb >= 256)
      hipLaunchKernelGGL((resample_bwd_coef_c16_dedup_kernel<1, 1, true>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else if (g_bwd_coef_variant == 4)
      hipLaunchKernelGGL((resample_bwd_coef_c16_kernel<6, 1>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    else
      hipLaunchKernelGGL((resample_bwd_coef_c16_kernel<8, 1>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, D, H, W, stp);
    int st2 = lf_launch_status();
    if (st2) return st2;
    hipLaunchKernelGGL(resample_bwd_coef_reduce, dim3(N), dim3(256), 0, s, partial, nblk, gcoef);
    return lf_launch_status();
  }
  if (vec)
    hipLaunchKernelGGL((resample_bwd_coef_kernel<4>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, N, D, H, W, C, lpv, make_steps(D, H, W));
  else
    hipLaunchKernelGGL((resample_bwd_coef_kernel<1>), grid, block, 0, s, gout, vol, bstride, coef, partial, nblk, vpb, bt, N, D, H, W, C, lpv, make_steps(D, H, W));
  int st = lf_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(resample_bwd_coef_reduce, dim3(N), dim3(256), 0, s, partial, nblk, gcoef);
  return lf_launch_status();
}

extern "C" int lf_resample3d_bwd_vol(const float* gout, const float* coef, int kind, float* gvol, int vol_n,
                                     int N, int D, int H, int W, int C, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return LF_EINVAL;
  if (vol_n != 1 && vol_n != N) return LF_EINVAL;
  const long bstride = vol_n == 1 ? 0 : (long)D * H * W * C;
  const long items = (long)D * H * W * C;
  dim3 grid((unsigned)min((items + 255) / 256, (long)65535 * 16), N), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (kind == LF_MAP_O2C)
    hipLaunchKernelGGL((resample_bwd_vol_kernel<LF_MAP_O2C>), grid, block, 0, s, gout, coef, gvol, bstride, N, D, H, W, C, make_steps(D, H, W));
  else if (kind == LF_MAP_C2O)
    hipLaunchKernelGGL((resample_bwd_vol_kernel<LF_MAP_C2O>), grid, block, 0, s, gout, coef, gvol, bstride, N, D, H, W, C, make_steps(D, H, W));
  else
    return LF_EINVAL;
  return lf_launch_status();
}

#ifdef STAGE_TS
extern "C" int lf_debug_stage_ts(void* dst) {                      // experimental builds only (tools/stage_timeline.py)
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_stage_ts), sizeof(unsigned long long) * 256 * 16, 0, hipMemcpyDeviceToHost);
}
#endif

// Tuning / A-B switch (not part of the functional interface): key 1 = resampler variant (1 generic, 2 lean, 3 lean + 16-channel gather).
// Returns the previous value, or LF_EINVAL for an unknown key.
extern "C" int lf_set_tuning(int key, int value) {
  if (key == 1) {
    const int prev = g_resample_variant;
    if (value >= 1 && value <= 5) g_resample_variant = value;
    return prev;
  }
  if (key == 4) {
    const int prev = g_splat_variant;
    if (value >= 1 && value <= 4) g_splat_variant = value;
    return prev;
  }
  if (key == 6) {
    const int prev = g_splat_chunk_cap;
    if (value >= 0) g_splat_chunk_cap = value;
    return prev;
  }
  if (key == 3) return lf_internal_fused_set_cfg(value);            // fused wide-conv GEMM: workgroup shape 0..3, -1 = by shape
  if (key == 5) return lf_internal_ring_bf16_set_wgs(value);        // bf16 ring convolution: resident workgroups per CU
  if (key == 2) {
    const int prev = g_bwd_coef_variant;
    if (value >= 1 && value <= 11) g_bwd_coef_variant = value;
    return prev;
  }
  return LF_EINVAL;
}

// Deterministic form of lf_resample3d_bwd_vol (no float atomics): see resample_bwd_vol_fixed_kernel.  gvol is
// overwritten (no zero-initialisation needed).  scratch: lf_resample3d_bwd_vol_det_scratch_bytes(...) bytes.
extern "C" size_t lf_resample3d_bwd_vol_det_scratch_bytes(int vol_n, int D, int H, int W, int C) {
  if (vol_n <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return 0;
  return (size_t)vol_n * D * H * W * C * sizeof(long long) + 256;
}

extern "C" int lf_resample3d_bwd_vol_det(const float* gout, const float* coef, int kind, float* gvol, int vol_n, void* scratch,
                                         size_t scratch_bytes, int N, int D, int H, int W, int C, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || C <= 0) return LF_EINVAL;
  if ((vol_n != 1 && vol_n != N) || (kind != LF_MAP_O2C && kind != LF_MAP_C2O)) return LF_EINVAL;
  const long items = (long)D * H * W * C, total = items * vol_n;
  if (scratch == nullptr || scratch_bytes < lf_resample3d_bwd_vol_det_scratch_bytes(vol_n, D, H, W, C)) return LF_ENOSPC;
  if ((((uintptr_t)scratch) & 7u) != 0) return LF_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long ng = items * N;
  const int nbx = (W + 3) / 4, nby = (H + 3) / 4, nbz = (D + 3) / 4;
  const long nblk = (long)nbx * nby * nbz;
  const int ntx = (W + STX - 1) / STX, nty = (H + STY - 1) / STY, ntz = (D + STZ - 1) / STZ;
  const int nsx = (nbx + 3) / 4, nsy = (nby + 3) / 4, nsz = (nbz + 3) / 4, nsb = nsx * nsy * nsz;
  if (g_splat_variant >= 2 && C == 16 && lf_aligned16(gout) && lf_aligned16(gvol) && D < 0x7fff && H < 0x7fff && W < 0x7fff &&
      nblk < 0x7fffffffL / 4 && (long)ntx * nty * ntz < 0x7fffffffL && N <= 65535 &&
      scratch_bytes >= 256 + (size_t)N * (nblk + nsb) * sizeof(uint3)) {
    // tiled form: scratch = [amax (256 B)] [block boxes: N x blocks x 12 B] [super-block boxes: N x super-blocks x 12 B]
    unsigned* amax = (unsigned*)scratch;
    uint3* bbox = (uint3*)((char*)scratch + 256);
    uint3* sbox = bbox + (size_t)N * nblk;
    hipError_t e = hipMemsetAsync(scratch, 0, 256, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)min((ng + 255) / 256, 4096L)), dim3(256), 0, s, gout, ng, amax);
    const Steps stp = make_steps(D, H, W);
    const dim3 gb((unsigned)((nblk + 3) / 4), (unsigned)N), gs2((unsigned)((nsb + 3) / 4), (unsigned)N),
        gt((unsigned)((long)ntx * nty * ntz), (unsigned)vol_n);
    if (kind == LF_MAP_O2C)
      hipLaunchKernelGGL((splat_bbox_kernel<LF_MAP_O2C>), gb, dim3(256), 0, s, coef, bbox, (int)nblk, nbx, nby, D, H, W, stp);
    else
      hipLaunchKernelGGL((splat_bbox_kernel<LF_MAP_C2O>), gb, dim3(256), 0, s, coef, bbox, (int)nblk, nbx, nby, D, H, W, stp);
    hipLaunchKernelGGL(splat_bbox2_kernel, gs2, dim3(256), 0, s, bbox, sbox, (int)nblk, nsb, nbx, nby, nbz, nsx, nsy);
    if (kind == LF_MAP_O2C)
      hipLaunchKernelGGL((splat_tile_kernel<LF_MAP_O2C>), gt, dim3(256), 0, s, gout, coef, bbox, sbox, amax, gvol, vol_n, N, (int)nblk, nsb,
                         nbx, nby, nbz, nsx, nsy, ntx, nty, D, H, W, stp);
    else
      hipLaunchKernelGGL((splat_tile_kernel<LF_MAP_C2O>), gt, dim3(256), 0, s, gout, coef, bbox, sbox, amax, gvol, vol_n, N, (int)nblk, nsb,
                         nbx, nby, nbz, nsx, nsy, ntx, nty, D, H, W, stp);
    return lf_launch_status();
  }
  unsigned long long* acc = (unsigned long long*)scratch;
  unsigned* amax = (unsigned*)((char*)scratch + (size_t)total * sizeof(long long));
  hipError_t e = hipMemsetAsync(scratch, 0, (size_t)total * sizeof(long long) + 256, s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)min((ng + 255) / 256, 4096L)), dim3(256), 0, s, gout, ng, amax);
  int st = lf_launch_status();
  if (st) return st;
  const long bstride = vol_n == 1 ? 0 : items;
  dim3 grid((unsigned)min((items + 255) / 256, (long)65535 * 16), N), block(256);
  if (kind == LF_MAP_O2C)
    hipLaunchKernelGGL((resample_bwd_vol_fixed_kernel<LF_MAP_O2C>), grid, block, 0, s, gout, coef, acc, bstride, amax, N, D, H, W, C, make_steps(D, H, W));
  else
    hipLaunchKernelGGL((resample_bwd_vol_fixed_kernel<LF_MAP_C2O>), grid, block, 0, s, gout, coef, acc, bstride, amax, N, D, H, W, C, make_steps(D, H, W));
  st = lf_launch_status();
  if (st) return st;
  hipLaunchKernelGGL(fixed_to_float_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const long long*)acc, amax, gvol, total);
  return lf_launch_status();
}

// ---- storage-type variants for the training step's bf16 storage policy (16-channel volumes only) ----
extern "C" int lf_resample3d_fwd_io(const void* vol, int vol_n, const float* coef, int kind, void* out,
                                    int N, int D, int H, int W, int io, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || io < 0 || io > 3) return LF_EINVAL;
  if ((vol_n != 1 && vol_n != N) || (kind != LF_MAP_O2C && kind != LF_MAP_C2O)) return LF_EINVAL;
  if (!lf_aligned16(vol) || !lf_aligned16(out)) return LF_EALIGN;
  if ((long)D * H * W * 64 >= 0xffffffffL) return LF_EINVAL;
  const long bstride = vol_n == 1 ? 0 : (long)D * H * W * 16;
  const int nbz4 = (D + 3) >> 2;
  if ((long)nbz4 * N > 65535 || ((H + 3) >> 2) > 65535) return LF_EINVAL;
  dim3 g4((unsigned)((W + 3) >> 2), (unsigned)((H + 3) >> 2), (unsigned)(nbz4 * N)), block(256);
  const Steps st = make_steps(D, H, W);
  typedef void (*kern_t)(const float*, long, const float*, float*, int, int, int, int, Steps);
  static const kern_t kerns[2][4] = {
      {resample_fwd_c16_kernel<LF_MAP_O2C, 0>, resample_fwd_c16_kernel<LF_MAP_O2C, 1>, resample_fwd_c16_kernel<LF_MAP_O2C, 2>, resample_fwd_c16_kernel<LF_MAP_O2C, 3>},
      {resample_fwd_c16_kernel<LF_MAP_C2O, 0>, resample_fwd_c16_kernel<LF_MAP_C2O, 1>, resample_fwd_c16_kernel<LF_MAP_C2O, 2>, resample_fwd_c16_kernel<LF_MAP_C2O, 3>}};
  hipLaunchKernelGGL(kerns[kind == LF_MAP_O2C ? 0 : 1][io], g4, block, 0, (hipStream_t)stream, (const float*)vol, bstride, coef, (float*)out, D, H, W, nbz4, st);
  return lf_launch_status();
}

// scratch: the tile form: [amax (256 B)] [block boxes] [super-block boxes] (a few MB); the binned form: [amax (256 B)]
// [counters | cursors | offsets: 3 x chunk x tiles u32] [lists: chunk x voxels x 8 u32], chunk = samples per pass (lists of at
// most 512 MB)
static bool splat_binned_ok(int vol_n, int N, int D, int H, int W) {
  if (g_splat_variant == 3 || W > 1024 || H > 1024 || D > 4096) return false;
  // a volume per sample: passes of `chunk` samples; one shared volume: its tile accumulators live for ONE pass, all samples in it
  return vol_n == N || splat_bin_chunk(N, (long)D * H * W) == N;
}

extern "C" size_t lf_resample3d_bwd_vol_det_io_scratch_bytes(int vol_n, int N, int D, int H, int W) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
  const long nbx = (W + 3) / 4, nby = (H + 3) / 4, nbz = (D + 3) / 4;
  const long nsb = ((nbx + 3) / 4) * ((nby + 3) / 4) * ((nbz + 3) / 4);
  const size_t boxes = 256 + (size_t)N * (size_t)(nbx * nby * nbz + nsb) * sizeof(uint3);
  const long nvox = (long)D * H * W;
  const long nt = (long)((W + STX - 1) / STX) * ((H + STY - 1) / STY) * ((D + STZ - 1) / STZ);
  const long cv = splat_bin_chunk(N, nvox);
  const size_t lists = 256 + (size_t)(3 * cv * nt * 4) + (size_t)(cv * nvox * 32);
  // (sized for either form: lf_set_tuning may switch between them after the caller asked)
  return boxes > lists ? boxes : lists;
}

extern "C" int lf_resample3d_bwd_vol_det_io(const void* gout, const float* coef, int kind, void* gvol, int vol_n, void* scratch,
                                            size_t scratch_bytes, int N, int D, int H, int W, int io, void* stream) {
  lf_clear_error();
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || io < 0 || io > 3) return LF_EINVAL;
  if ((vol_n != 1 && vol_n != N) || (kind != LF_MAP_O2C && kind != LF_MAP_C2O)) return LF_EINVAL;
  if (scratch == nullptr || scratch_bytes < lf_resample3d_bwd_vol_det_io_scratch_bytes(vol_n, N, D, H, W)) return LF_ENOSPC;
  if ((((uintptr_t)scratch) & 7u) != 0 || !lf_aligned16(gout) || !lf_aligned16(gvol)) return LF_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  const long ng = (long)D * H * W * 16 * N;
  if (splat_binned_ok(vol_n, N, D, H, W)) {
    // binned form (list entries pack z | y | x into 12 + 10 + 10 bits)
    const long nvox = (long)D * H * W;
    const int ntx = (W + STX - 1) / STX, nty = (H + STY - 1) / STY, ntz = (D + STZ - 1) / STZ;
    const long nt = (long)ntx * nty * ntz;
    if (nvox >= 0xffffffffL || nt >= 0x7fffffffL || N > 65535) return LF_EINVAL;
    const int cv = splat_bin_chunk(N, nvox);
    const long cap = nvox * 8;
    unsigned* amax = (unsigned*)scratch;
    unsigned* cnt = (unsigned*)((char*)scratch + 256);
    unsigned* cur = cnt + (size_t)cv * nt;
    unsigned* off = cur + (size_t)cv * nt;
    unsigned* list = off + (size_t)cv * nt;
    hipError_t e = hipMemsetAsync(scratch, 0, 256, s);
    if (e != hipSuccess) return (int)e;
    if (io & 1) hipLaunchKernelGGL(absmax_bf16_kernel, dim3((unsigned)min((ng + 255) / 256, 4096L)), dim3(256), 0, s, (const __bf16*)gout, ng, amax);
    else hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)min((ng + 255) / 256, 4096L)), dim3(256), 0, s, (const float*)gout, ng, amax);
    const Steps stp = make_steps(D, H, W);
    typedef void (*tile_t)(const float*, const float*, const unsigned*, const unsigned*, const unsigned*, const unsigned*, float*, int, int,
                           int, long, long, int, int, int, int, int, int, Steps);
    static const tile_t tiles[2][4] = {
        {splat_binned_tile_kernel<LF_MAP_O2C, 0>, splat_binned_tile_kernel<LF_MAP_O2C, 1>, splat_binned_tile_kernel<LF_MAP_O2C, 2>,
         splat_binned_tile_kernel<LF_MAP_O2C, 3>},
        {splat_binned_tile_kernel<LF_MAP_C2O, 0>, splat_binned_tile_kernel<LF_MAP_C2O, 1>, splat_binned_tile_kernel<LF_MAP_C2O, 2>,
         splat_binned_tile_kernel<LF_MAP_C2O, 3>}};
    static const tile_t tiles16[2][4] = {
        {splat_binned_tile16_kernel<LF_MAP_O2C, 0>, splat_binned_tile16_kernel<LF_MAP_O2C, 1>, splat_binned_tile16_kernel<LF_MAP_O2C, 2>,
         splat_binned_tile16_kernel<LF_MAP_O2C, 3>},
        {splat_binned_tile16_kernel<LF_MAP_C2O, 0>, splat_binned_tile16_kernel<LF_MAP_C2O, 1>, splat_binned_tile16_kernel<LF_MAP_C2O, 2>,
         splat_binned_tile16_kernel<LF_MAP_C2O, 3>}};
    for (int n0 = 0; n0 < N; n0 += cv) {
      const int m = min(cv, N - n0);
      e = hipMemsetAsync(cnt, 0, (size_t)2 * cv * nt * 4, s);
      if (e != hipSuccess) return (int)e;
      const dim3 gbin((unsigned)(((long)((W + 3) / 4) * ((H + 3) / 4) * ((D + 3) / 4) + 3) / 4), (unsigned)m);
      if (kind == LF_MAP_O2C) {
        hipLaunchKernelGGL((splat_bin_kernel<LF_MAP_O2C, false>), gbin, dim3(256), 0, s, coef, cnt, off, list, n0, nvox, cap, (int)nt, ntx, nty, D, H, W, stp);
        hipLaunchKernelGGL(splat_scan_kernel, dim3((unsigned)m), dim3(256), 0, s, cnt, off, (int)nt);
        hipLaunchKernelGGL((splat_bin_kernel<LF_MAP_O2C, true>), gbin, dim3(256), 0, s, coef, cur, off, list, n0, nvox, cap, (int)nt, ntx, nty, D, H, W, stp);
      } else {
        hipLaunchKernelGGL((splat_bin_kernel<LF_MAP_C2O, false>), gbin, dim3(256), 0, s, coef, cnt, off, list, n0, nvox, cap, (int)nt, ntx, nty, D, H, W, stp);
        hipLaunchKernelGGL(splat_scan_kernel, dim3((unsigned)m), dim3(256), 0, s, cnt, off, (int)nt);
        hipLaunchKernelGGL((splat_bin_kernel<LF_MAP_C2O, true>), gbin, dim3(256), 0, s, coef, cur, off, list, n0, nvox, cap, (int)nt, ntx, nty, D, H, W, stp);
      }
      const int shared = vol_n == 1 && N > 1;
      hipLaunchKernelGGL((g_splat_variant == 4 ? tiles16 : tiles)[kind == LF_MAP_O2C ? 0 : 1][io], dim3((unsigned)nt, (unsigned)(shared ? 1 : m)), dim3(g_splat_variant == 4 ? 256 : SBT_THREADS), 0, s, (const float*)gout,
                         coef, cnt, off, list, amax, (float*)gvol, n0, m, shared, nvox, cap, (int)nt, ntx, nty, D, H, W, stp);
    }
    return lf_launch_status();
  }
  const int nbx = (W + 3) / 4, nby = (H + 3) / 4, nbz = (D + 3) / 4;
  const long nblk = (long)nbx * nby * nbz;
  const int ntx = (W + STX - 1) / STX, nty = (H + STY - 1) / STY, ntz = (D + STZ - 1) / STZ;
  const int nsx = (nbx + 3) / 4, nsy = (nby + 3) / 4, nsz = (nbz + 3) / 4, nsb = nsx * nsy * nsz;
  if (!(D < 0x7fff && H < 0x7fff && W < 0x7fff && nblk < 0x7fffffffL / 4 && (long)ntx * nty * ntz < 0x7fffffffL && N <= 65535 &&
        scratch_bytes >= 256 + (size_t)N * (nblk + nsb) * sizeof(uint3))) return LF_EINVAL;
  unsigned* amax = (unsigned*)scratch;
  uint3* bbox = (uint3*)((char*)scratch + 256);
  uint3* sbox = bbox + (size_t)N * nblk;
  hipError_t e = hipMemsetAsync(scratch, 0, 256, s);
  if (e != hipSuccess) return (int)e;
  if (io & 1) hipLaunchKernelGGL(absmax_bf16_kernel, dim3((unsigned)min((ng + 255) / 256, 4096L)), dim3(256), 0, s, (const __bf16*)gout, ng, amax);
  else hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)min((ng + 255) / 256, 4096L)), dim3(256), 0, s, (const float*)gout, ng, amax);
  const Steps stp = make_steps(D, H, W);
  const dim3 gb((unsigned)((nblk + 3) / 4), (unsigned)N), gs2((unsigned)((nsb + 3) / 4), (unsigned)N),
      gt((unsigned)((long)ntx * nty * ntz), (unsigned)vol_n);
  if (kind == LF_MAP_O2C)
    hipLaunchKernelGGL((splat_bbox_kernel<LF_MAP_O2C>), gb, dim3(256), 0, s, coef, bbox, (int)nblk, nbx, nby, D, H, W, stp);
  else
    hipLaunchKernelGGL((splat_bbox_kernel<LF_MAP_C2O>), gb, dim3(256), 0, s, coef, bbox, (int)nblk, nbx, nby, D, H, W, stp);
  hipLaunchKernelGGL(splat_bbox2_kernel, gs2, dim3(256), 0, s, bbox, sbox, (int)nblk, nsb, nbx, nby, nbz, nsx, nsy);
  typedef void (*kern_t)(const float*, const float*, const uint3*, const uint3*, const unsigned*, float*, int, int, int, int, int, int, int,
                         int, int, int, int, int, int, int, Steps);
  static const kern_t kerns[2][4] = {
      {splat_tile_kernel<LF_MAP_O2C, 0>, splat_tile_kernel<LF_MAP_O2C, 1>, splat_tile_kernel<LF_MAP_O2C, 2>, splat_tile_kernel<LF_MAP_O2C, 3>},
      {splat_tile_kernel<LF_MAP_C2O, 0>, splat_tile_kernel<LF_MAP_C2O, 1>, splat_tile_kernel<LF_MAP_C2O, 2>, splat_tile_kernel<LF_MAP_C2O, 3>}};
  hipLaunchKernelGGL(kerns[kind == LF_MAP_O2C ? 0 : 1][io], gt, dim3(256), 0, s, (const float*)gout, coef, bbox, sbox, amax, (float*)gvol, vol_n, N,
                     (int)nblk, nsb, nbx, nby, nbz, nsx, nsy, ntx, nty, D, H, W, stp);
  return lf_launch_status();
}
