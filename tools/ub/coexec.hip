// Do MFMAs (fp32 16x16x4 / f16 16x16x32) and VALU ops overlap -- from two waves on one SIMD, and within one wave?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void __launch_bounds__(512) k(float* out, int iters, int modeA, int modeB) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: no exec-mask code around the loops)
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
  else if (mode >= 5 && mode <= 7) {   // f16 MFMA 16x16x32, 4 independent chains (+ 8 / 4 v_add_f32 in the SAME wave: modes 6 / 7)
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    f16x8 x, w;
    for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 1e-3f + e); w[e] = (_Float16)1.0f; }
    float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7, c = 1.5f;
#define M4(V) a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a0, 0, 0, 0); V(v0, v1) a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a1, 0, 0, 0); V(v2, v3) \
              a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a2, 0, 0, 0); V(v4, v5) a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(w, x, a3, 0, 0, 0); V(v6, v7)
#define V0(p, q)
#define V2(p, q) asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(p), "+v"(q) : "v"(c));
#define V1(p, q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(p) : "v"(c));
    if (mode == 5) for (int i = 0; i < iters; ++i) { M4(V0) }
    else if (mode == 6) for (int i = 0; i < iters; ++i) { M4(V2) }
    else for (int i = 0; i < iters; ++i) { M4(V1) }
    r = a0[0] + a1[1] + a2[2] + a3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  else if (mode >= 8) {          // 8 per iteration of one more VALU opcode, independent registers
    float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7, c = 1.5f;
#define OP8(T) for (int i = 0; i < iters; ++i) asm volatile(T("%0") T("%1") T("%2") T("%3") T("%4") T("%5") T("%6") T("%7") \
      : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c));
#define T_MIXLO(r) "v_fma_mixlo_f16 " r ", " r ", %8, 0\n"
#define T_MIXSRC(r) "v_fma_mixlo_f16 " r ", %8, %8, -" r " op_sel_hi:[0,0,1]\n"
#define T_CVTPK(r) "v_cvt_pk_f16_f32 " r ", " r ", %8\n"
#define T_CVTF32(r) "v_cvt_f32_f16 " r ", " r "\n"
#define T_FMA(r) "v_fma_f32 " r ", " r ", %8, %8\n"
#define T_CND(r) "v_cndmask_b32 " r ", " r ", %8, vcc\n"
#define T_RSQ(r) "v_rsq_f32 " r ", " r "\n"
#define T_PERM(r) "v_permlane32_swap_b32 " r ", %8\n"
#define T_PKMUL(r) "v_max_f32 " r ", " r ", %8\n"
    if (mode == 8) { OP8(T_MIXLO) } else if (mode == 9) { OP8(T_MIXSRC) } else if (mode == 10) { OP8(T_CVTPK) } else if (mode == 11) { OP8(T_CVTF32) }
    else if (mode == 12) { OP8(T_FMA) } else if (mode == 13) { OP8(T_CND) } else if (mode == 14) { OP8(T_RSQ) } else if (mode == 15) { OP8(T_PKMUL) }
    r = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
  }
  if (r == 123.456f) out[threadIdx.x] = r;
}
int main() {
  float* out; hipMalloc(&out, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  const char* names[] = {"idle", "mfma_f32(4/iter)", "v_add_f32(8/iter)", "v_pk_add_f32(8/iter)", "v_add_u32(8/iter)", "mfma_f16_16x16x32(4/iter)", "mfma_f16(4)+v_add_f32(8) same wave", "mfma_f16(4)+v_add_f32(4) same wave",
                         "v_fma_mixlo_f16 (f32 src)", "v_fma_mixlo_f16 (f16 src)", "v_cvt_pk_f16_f32", "v_cvt_f32_f16", "v_fma_f32", "v_cndmask_b32", "v_rsq_f32", "v_max_f32"};
  int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 2}, {1, 3}, {1, 4}, {5, 0}, {5, 5}, {5, 2}, {5, 3}, {5, 4}, {6, 0}, {6, 6}, {7, 0}, {7, 7}, {8, 0}, {5, 8}, {9, 0}, {5, 9}, {10, 0}, {5, 10}, {11, 0}, {5, 11}, {12, 0}, {5, 12}, {13, 0}, {5, 13}, {14, 0}, {5, 14}, {15, 0}, {5, 15}};
  for (auto& c : combos) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, c[0], c[1]);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, c[0], c[1]);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("waves0-3: %-36s waves4-7: %-36s %8.3f ms  (%.1f clk/iter @2.4GHz)\n", names[c[0]], names[c[1]], ms, ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
