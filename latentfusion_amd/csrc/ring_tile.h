// The tile / ring geometry and the small device helpers shared by the ring convolutions (conv_split.hip: the f16x3 and bf16
// forms of the 16 -> 16 convolution; conv_gru.hip: the multi-output forms of the ConvGRU recurrence).  See conv_split.hip for
// the organisation these constants describe.
#pragma once
#include "lf_common.h"

typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4s __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef unsigned u32x2s __attribute__((ext_vector_type(2)));

namespace {

constexpr int TXs = 16, TYs = 8, TZs = 2, HXs = 18, HYs = 10, HZs = TZs + 2;
constexpr int RYs = 4;                                   // output rows per wave
constexpr int RINGs = 6;                                 // z-plane slots: 4 being read + 2 being filled
constexpr int PLANE_VOX = HYs * HXs;                     // 180 voxels
constexpr int PLANE_B = PLANE_VOX * 32;                  // 5,760 B per plane and kind (hi or lo)
constexpr int LO_OFF = RINGs * PLANE_B;                  // 34,560: the lo planes follow the six hi plane slots
constexpr int GUARD_B = 576;                             // operand R reads up to 18 voxels past its plane
constexpr int LDSs = 2 * LO_OFF + GUARD_B;               // 69,696 B
constexpr int NPAIR = 14;
constexpr int NPIECE = 6;                                // 1 KiB buffer loads per wave and half plane: 5 rows + 1 (x = 16, 17)
constexpr int NOP = 5 * (RYs + 2);                       // B operands per wave and tile: P[3][6], Q[6], R[6]

// tap index = kz*9 + ky*3 + kx
__host__ __device__ constexpr int pair_first(int p) {
  return p < 9 ? (p % 3) * 3 + (p / 3) : (p < 12 ? 18 + (p - 9) * 3 : (p == 12 ? 20 : 26));
}
__host__ __device__ constexpr int pair_second(int p) {
  return p < 9 ? pair_first(p) + 9 : (p < 12 ? pair_first(p) + 1 : (p == 12 ? 23 : -1));
}

template <int I> struct IC { static constexpr int v = I; };
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (B < E) { f(IC<B>{}); static_for<B + 1, E>(f); }
}
// operand i: 0..17 = P[kx = i / 6][h = i % 6], 18..23 = Q[h], 24..29 = R[h]
__host__ __device__ constexpr int op_cls(int i) { return i < 18 ? 0 : (i < 24 ? 1 : 2); }
__host__ __device__ constexpr int op_h(int i) { return i % 6; }
__host__ __device__ constexpr int op_kx(int i) { return i < 18 ? i / 6 : 0; }

__device__ __forceinline__ void lds_barrier_s() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// sum over the four lanes n, n+16, n+32, n+48 (the four channel quarters of a voxel), same value in all four:
// v_permlane32_swap / v_permlane16_swap exchange half-waves / neighbouring rows of 16 in the VALU (a __shfl_xor is a
// ds_bpermute: an LDS round trip queued behind the operand reads of every wave on the CU)
__device__ __forceinline__ float quarter_sum(float v) {
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// v_rsq_f32 / v_rcp_f32 (1 ulp) + one Newton step
__device__ __forceinline__ float fast_rsqrt_s(float x) {
  const float r = __builtin_amdgcn_rsqf(x);
  return r * (1.5f - 0.5f * x * r * r);
}
__device__ __forceinline__ float fast_rcp_s(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return r * (2.f - x * r);
}
__device__ __forceinline__ f32x4 mfma_k32(const f16x8 a, const f16x8 b, const f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma_k32(const bf16x8s a, const bf16x8s b, const f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float rb16(float v) { return (float)(__bf16)v; }
__device__ __forceinline__ int mod6(int v) { return v >= 6 ? v - 6 : v; }      // v in [0, 12)


}  // namespace
