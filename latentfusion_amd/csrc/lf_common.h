// Shared device/host helpers for the gfx950 kernels of liblf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lf_hip.h"

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

__device__ __forceinline__ float lf_wave_sum(float v) {
  // fixed-order butterfly over the 64 lanes of a wavefront (deterministic)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float lf_lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }
