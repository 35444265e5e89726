// Shared device/host helpers for the gfx950 kernels of liblf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lf_hip.h"
#include "../../include/lf_hip_experimental.h"

#define LF_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

// hipGetLastError() is sticky per thread and shared with every other HIP user in the process (the host
// framework included): each entry point clears it first so that lf_launch_status() reports only its own launch.
static inline void lf_clear_error() { (void)hipGetLastError(); }

static inline int lf_launch_status() {
  hipError_t e = hipGetLastError();
  return (int)e;
}

static inline bool lf_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, DEVICE): `done` keeps one bit per device ordinal, so a
// process that drives several GPUs (one host thread each, e.g. autograd's per-device backward threads) sets it on each of
// them, and concurrent first launches at worst set it twice.
#include <atomic>
typedef std::atomic<unsigned long long> lf_devmask_t;
static inline hipError_t lf_ensure_dyn_lds(lf_devmask_t& done, const void* fn, int bytes) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

__device__ __forceinline__ float lf_wave_sum(float v) {
  // fixed-order butterfly over the 64 lanes of a wavefront (deterministic)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Max-abs side channel between kernels (power-of-two pre-scaling of split-precision inputs): LF_AMAX_SLOTS
// partial maxima, one per 128-byte line, so that the producers' atomic maxima spread over the L2 channels
// instead of serialising on one address (131k waves on one address cost ~0.7 ms).  The buffer is
// LF_AMAX_FLOATS floats, zero-initialised by the caller; non-negative floats order like unsigned ints.
#define LF_AMAX_SLOTS 64
#define LF_AMAX_STRIDE 32
static_assert(LF_AMAX_FLOATS == LF_AMAX_SLOTS * LF_AMAX_STRIDE, "lf_hip.h: LF_AMAX_FLOATS");

__device__ __forceinline__ void lf_amax_publish(float* base, float wave_max, int lane) {
  if (lane == 0 && wave_max > 0.f)
    atomicMax((unsigned int*)(base + (blockIdx.x % LF_AMAX_SLOTS) * LF_AMAX_STRIDE), __float_as_uint(wave_max));
}

// whole-wave read: every lane returns the maximum over the slots
__device__ __forceinline__ float lf_amax_read(const float* base, int lane) {
  float m = base[(lane % LF_AMAX_SLOTS) * LF_AMAX_STRIDE];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  return m;
}

__device__ __forceinline__ float lf_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// A/B switches that live in other translation units (lf_set_tuning dispatches to them)
int lf_internal_ring_bf16_set_wgs(int v);                         // conv_split.hip: workgroups per CU of lf_conv3d_c16_ring_bf16 (2 or 3)
int lf_internal_fused_set_cfg(int v);                             // wino_fused.hip: workgroup shape of the fused GEMM, -1 = by shape
