// LDS atomic throughput on gfx950: N waves x ITER atomic adds per lane, addresses spread like the splat's
// (lane -> its own accumulator group, groups of 4 lanes = 4 adjacent 4-element runs of one 16-element record).
//   hipcc --offload-arch=gfx950 -O3 tools/ub/lds_atomic.hip -o /tmp/lds_atomic && /tmp/lds_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>

template <typename T>
__global__ void __launch_bounds__(256) k(unsigned* out, int iters, int stride) {
  __shared__ T acc[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = (T)0;
  __syncthreads();
  const int lane = threadIdx.x;
  const int q = lane & 3, v = lane >> 2;                       // 64 voxels x 4 quads
  long t0 = clock64();
  int base = (v * stride) & 255;                               // record of 16 elements
  for (int i = 0; i < iters; ++i) {
    T* d = acc + base * 16 + q * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(d + e, (T)(i + e + 1));
    base = (base + 37) & 255;
  }
  __syncthreads();
  long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (unsigned)(t1 - t0);
  if (threadIdx.x == 1 && acc[5] == (T)123456789) out[0] = 0;
}

template <typename T>
void run(const char* name, int wgs_per_cu) {
  unsigned* d;
  hipMalloc(&d, 4096 * 4);
  const int iters = 2048, blocks = 256 * wgs_per_cu;
  for (int stride = 1; stride <= 1; ++stride) {
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, iters, stride);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, iters, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double lane_atomics = (double)blocks * 256 * iters * 4;
    printf("%-6s wgs/cu %d: %.3f ms, %.2f lane-atomics per ns per CU-ish (%.1f per clk @2.1GHz per CU)\n", name, wgs_per_cu, ms,
           lane_atomics / (ms * 1e6) / 256, lane_atomics / (ms * 1e6) / 256 / 2.1);
  }
  hipFree(d);
}

int main() {
  for (int w : {1, 4}) {
    run<unsigned long long>("u64", w);
    run<unsigned>("u32", w);
    run<float>("f32", w);
    run<double>("f64", w);
  }
  return 0;
}
