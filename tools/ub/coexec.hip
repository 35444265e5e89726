// Do f32 MFMAs and fp32 VALU ops from two waves on one SIMD overlap?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(512) k(float* out, int iters, int modeA, int modeB) {
  const int wave = threadIdx.x >> 6;
  const int mode = (wave < 4) ? modeA : modeB;      // waves w and w+4 share a SIMD
  float r = 0.f;
  if (mode == 1) {             // MFMA f32 16x16x4, 4 independent chains
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, w = 1.0001f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, x, a3, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3];
  } else if (mode == 2) {      // scalar fp32 adds, 8 independent chains, 8 per iteration
    float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7, c = 1.5f;
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                   "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c));
    }
    r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  } else if (mode == 3) {      // packed fp32 adds: 8 v_pk_add_f32 per iteration (16 floats)
    f32x2 v0 = {1, 2}, v1 = v0, v2 = v0, v3 = v0, v4 = v0, v5 = v0, v6 = v0, v7 = v0, c = {1.5f, 2.5f};
    v0[0] = threadIdx.x;
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                   "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c));
    }
    r = v0[0] + v1[1] + v2[0] + v3[1] + v4[0] + v5[1] + v6[0] + v7[1];
  } else if (mode == 4) {      // integer VALU
    int v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7, c = 3;
    for (int i = 0; i < iters; ++i) {
      asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                   "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                   : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c));
    }
    r = (float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7);
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}
int main() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[] = {"idle", "mfma_f32(4/iter)", "v_add_f32(8/iter)", "v_pk_add_f32(8/iter)", "v_add_u32(8/iter)"};
  int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 2}, {1, 3}, {1, 4}};
  for (auto& c : combos) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, c[0], c[1]);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, c[0], c[1]);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves0-3: %-22s waves4-7: %-22s %8.3f ms  (%.1f clk/iter @2.4GHz)\n", names[c[0]], names[c[1]], ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
