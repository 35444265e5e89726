// Microbenchmark for VERDICT r03 item 7: does the L1 of gfx950 serve the two x-adjacent taps of a trilinear gather from a
// channels-last volume (2 x 64-byte records = 128 contiguous bytes) as ONE look-up when eight lanes request them with one
// instruction, instead of two look-ups from two instructions of four lanes each?
//   A: lane = (voxel, quarter): 8 x buffer_load_b128 per lane, 16 voxels per wave instruction   (the shipped gather's shape)
//   B: lane = (voxel, x-corner, quarter): 4 x buffer_load_b128 per lane, 8 voxels per wave instruction, the two x-corner
//      lanes of a voxel combined with one DPP row shift per channel
// Same object->camera map, same trilinear arithmetic up to the association of the x sum; the host checks A == B to 1e-6 and
// times both; rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum gives the look-ups.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ub/gather128.hip -o /tmp/gather128 && /tmp/gather128
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Map { float c[18]; };

__device__ __forceinline__ void tap_axis(float g, int size, u32 stride, u32& o0, u32& d1, float& t) {
  float p = ((g + 1.f) * (float)size - 1.f) * 0.5f;
  p = fminf(fmaxf(p, 0.f), (float)(size - 1));
  const float f = floorf(p);
  t = p - f;
  const int i0 = (int)f;
  o0 = (u32)i0 * stride;
  d1 = (i0 + 1 <= size - 1) ? stride : 0u;
}

__device__ __forceinline__ void eval(const Map& m, int x, int y, int z, int S, u32& o000, u32& dx, u32& dy, u32& dz,
                                     float& tx, float& ty, float& tz) {
  const float st = 1.f / (float)(S - 1);
  const float a = x * st, b = y * st, k = z * st, ak = a * k, bk = b * k;
  const float gx = m.c[0] + m.c[3] * a + m.c[6] * b + m.c[9] * k + m.c[12] * ak + m.c[15] * bk;
  const float gy = m.c[1] + m.c[4] * a + m.c[7] * b + m.c[10] * k + m.c[13] * ak + m.c[16] * bk;
  const float gz = m.c[2] + m.c[5] * a + m.c[8] * b + m.c[11] * k + m.c[14] * ak + m.c[17] * bk;
  u32 ox, oy, oz;
  tap_axis(gx, S, 64u, ox, dx, tx);
  tap_axis(gy, S, 64u * S, oy, dy, ty);
  tap_axis(gz, S, 64u * S * S, oz, dz, tz);
  o000 = ox + oy + oz;
}

__device__ __forceinline__ f32x4 ld(__amdgpu_buffer_rsrc_t rs, u32 off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
}

__global__ void __launch_bounds__(256) gather_a(const float* vol, const Map* maps, float* out, int S) {
  const int n = blockIdx.z / (S / 4), bz = blockIdx.z % (S / 4);
  const int q = threadIdx.x & 3, vs = threadIdx.x >> 2;
  const int x = blockIdx.x * 4 + (vs & 3), y = blockIdx.y * 4 + ((vs >> 2) & 3), z = bz * 4 + (vs >> 4);
  const u32 bytes = (u32)S * S * S * 64u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)vol, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)n * S * S * S * 16), 0, bytes, 0x00020000);
  u32 o, dx, dy, dz;
  float tx, ty, tz;
  eval(maps[n], x, y, z, S, o, dx, dy, dz, tx, ty, tz);
  o += q * 16u;
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cz = 0; cz < 2; ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy) {
      const u32 b = o + (cz ? dz : 0u) + (cy ? dy : 0u);
      const float wzy = (cz ? tz : 1.f - tz) * (cy ? ty : 1.f - ty);
      const f32x4 v0 = ld(rs, b), v1 = ld(rs, b + dx);
      r += (v0 * (1.f - tx) + v1 * tx) * wzy;
    }
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), ro, (int)((u32)((z * S + y) * S + x) * 64u + q * 16u), 0, 2);
}

__global__ void __launch_bounds__(256) gather_b(const float* vol, const Map* maps, float* out, int S) {
  // 32 voxels per 256 threads: a 4 x 4 x 2 block; lane = (voxel, x-corner, quarter)
  const int n = blockIdx.z / (S / 2), bz = blockIdx.z % (S / 2);
  const int q = threadIdx.x & 3, xs = (threadIdx.x >> 2) & 1, vs = threadIdx.x >> 3;
  const int x = blockIdx.x * 4 + (vs & 3), y = blockIdx.y * 4 + ((vs >> 2) & 3), z = bz * 2 + (vs >> 4);
  const u32 bytes = (u32)S * S * S * 64u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)vol, 0, bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (long)n * S * S * S * 16), 0, bytes, 0x00020000);
  u32 o, dx, dy, dz;
  float tx, ty, tz;
  eval(maps[n], x, y, z, S, o, dx, dy, dz, tx, ty, tz);
  o += q * 16u + (xs ? dx : 0u);
  const float wx = xs ? tx : 1.f - tx;
  f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int cz = 0; cz < 2; ++cz)
#pragma unroll
    for (int cy = 0; cy < 2; ++cy) {
      const u32 b = o + (cz ? dz : 0u) + (cy ? dy : 0u);
      const float wzy = (cz ? tz : 1.f - tz) * (cy ? ty : 1.f - ty);
      r += ld(rs, b) * (wx * wzy);
    }
  // x-corner 1 lanes sit 4 lanes above their x-corner 0 partners: row_shl:4 (0x104) brings their value down
#pragma unroll
  for (int e = 0; e < 4; ++e)
    r[e] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(r[e]), 0x104, 0xF, 0xF, true));
  if (xs == 0)
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, r), ro, (int)((u32)((z * S + y) * S + x) * 64u + q * 16u), 0, 2);
}

int main(int argc, char** argv) {
  const int S = 128, N = 8;
  const size_t volf = (size_t)S * S * S * 16, outf = volf * N;
  std::vector<float> hv(volf);
  srand(1);
  for (auto& v : hv) v = (float)rand() / RAND_MAX - 0.5f;
  // object -> camera maps of the loop's kind: a rotation by a few tens of degrees about a tilted axis, scale ~0.9, small shift
  std::vector<Map> hm(N);
  for (int n = 0; n < N; ++n) {
    const float ang = 0.3f + 0.25f * n, ax = 0.3f, ay = 0.8f, az = sqrtf(1.f - ax * ax - ay * ay), c = cosf(ang), s = sinf(ang), t = 1 - c;
    const float R[3][3] = {{t * ax * ax + c, t * ax * ay - s * az, t * ax * az + s * ay},
                           {t * ax * ay + s * az, t * ay * ay + c, t * ay * az - s * ax},
                           {t * ax * az - s * ay, t * ay * az + s * ax, t * az * az + c}};
    Map m;
    for (int i = 0; i < 18; ++i) m.c[i] = 0.f;
    for (int r = 0; r < 3; ++r) {                                   // g = R (2 (a,b,k) - 1) * 0.9: affine part only (c12..c17 = 0.02)
      m.c[r] = -0.9f * (R[r][0] + R[r][1] + R[r][2]) + 0.01f * n;
      m.c[3 + r] = 1.8f * R[r][0];
      m.c[6 + r] = 1.8f * R[r][1];
      m.c[9 + r] = 1.8f * R[r][2];
      m.c[12 + r] = 0.02f;
      m.c[15 + r] = -0.02f;
    }
    hm[n] = m;
  }
  float *dv, *da, *db;
  Map* dm;
  hipMalloc(&dv, volf * 4); hipMalloc(&da, outf * 4); hipMalloc(&db, outf * 4); hipMalloc(&dm, N * sizeof(Map));
  hipMemcpy(dv, hv.data(), volf * 4, hipMemcpyHostToDevice);
  hipMemcpy(dm, hm.data(), N * sizeof(Map), hipMemcpyHostToDevice);
  const dim3 ga(S / 4, S / 4, N * (S / 4)), gb(S / 4, S / 4, N * (S / 2));
  auto run = [&](int which) {
    if (which == 0) hipLaunchKernelGGL(gather_a, ga, dim3(256), 0, 0, dv, dm, da, S);
    else hipLaunchKernelGGL(gather_b, gb, dim3(256), 0, 0, dv, dm, db, S);
  };
  run(0); run(1);
  hipDeviceSynchronize();
  std::vector<float> ha(1 << 20), hb(1 << 20);
  double md = 0, mx = 0;
  for (int chunk = 0; chunk < 8; ++chunk) {
    const size_t off = (outf / 8) * chunk + 12345;
    hipMemcpy(ha.data(), da + off, ha.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), db + off, hb.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < ha.size(); ++i) { md = fmax(md, fabs((double)ha[i] - hb[i])); mx = fmax(mx, fabs((double)ha[i])); }
  }
  printf("max |A - B| = %.3e (max |A| = %.3f)\n", md, mx);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 5; ++rep)
    for (int which = 0; which < 2; ++which) {
      hipEventRecord(e0);
      for (int i = 0; i < 5; ++i) run(which);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s: %.4f ms per launch (8 x 128^3 x 16 channels)\n", which ? "B (8 lanes, 128 B)" : "A (4 lanes,  64 B)", ms / 5);
    }
  return 0;
}
