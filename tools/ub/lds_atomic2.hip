// LDS 64-bit atomic throughput on gfx950 by access SHAPE (round 6, for the splat's accumulators): a wave instruction covers
// 64 / LPE records, LPE lanes per record with 16 / LPE consecutive 8-byte slots each (LPE = 4 is the splat's lane quad per list
// entry, LPE = 16 one lane per channel: 128 contiguous bytes per record and instruction).  Records are picked pseudo-randomly
// per entry and iteration from the 256 of a 4 x 8 x 8 tile (SAME = 0) or all lanes' entries hit one record (SAME = 1: a border
// pile-up); record stride SACC slots.
//   hipcc --offload-arch=gfx950 -O3 tools/ub/lds_atomic2.hip -o /tmp/lds_atomic2 && /tmp/lds_atomic2
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int LPE, int SACC, int SAME>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters) {
  __shared__ unsigned long long acc[256 * SACC];
  for (int i = threadIdx.x; i < 256 * SACC; i += 256) acc[i] = 0ull;
  __syncthreads();
  constexpr int CPL = 16 / LPE;
  const int entry = threadIdx.x / LPE, sub = threadIdx.x % LPE;
  unsigned h = (unsigned)entry * 2654435761u + blockIdx.x * 40503u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    const int rec = SAME ? (i & 255) : (int)(h >> 24);
    unsigned long long* d = acc + rec * SACC + sub * CPL;
#pragma unroll
    for (int e = 0; e < CPL; ++e) atomicAdd(d + e, (unsigned long long)(i + e + 1));
  }
  __syncthreads();
  if (threadIdx.x == 1 && acc[5] == 123456789ull) out[0] = 0;
}

template <int LPE, int SACC, int SAME>
void run() {
  unsigned* d;
  hipMalloc(&d, 4096 * 4);
  const int iters = 4096 * LPE / 16 * 4, blocks = 256 * 4;          // the same number of lane-atomics for every LPE
  hipLaunchKernelGGL((k<LPE, SACC, SAME>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<LPE, SACC, SAME>), dim3(blocks), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double lane_atomics = (double)blocks * 256 * iters * (16 / LPE);
  printf("lanes/record %2d  stride %2d  %s: %.3f ms, %.1f lane-atomics per clk and CU (@2.4 GHz, 256 CUs)\n", LPE, SACC,
         SAME ? "one record " : "random recs", ms, lane_atomics / (ms * 1e6) / 256 / 2.4);
  hipFree(d);
}

int main() {
  run<4, 16, 0>(); run<4, 17, 0>(); run<8, 16, 0>(); run<8, 17, 0>(); run<16, 16, 0>(); run<16, 17, 0>(); run<16, 18, 0>();
  run<4, 17, 1>(); run<8, 17, 1>(); run<16, 17, 1>();
  return 0;
}
