// microbenchmark: the chip's VALU wave-instruction issue ceiling (the `roofline.issue.peak` of bench.py), measured, not assumed.
// One 64-lane workgroup = one wavefront; 1..8 wavefronts per SIMD (256 CUs x 4 SIMDs); every wave runs v_fma_f32 on 8 independent
// accumulators (no dependent-issue stalls) or on ONE dependent chain.  Prints G wave-instructions/s for each occupancy.
// MI355X guide (MI355X_MICROARCH.md "Wave scheduling"): SIMD-32, one wave64 VALU instruction per 2 cycles -> 1024 SIMDs x 2.4 GHz / 2 = 1228.8 G/s.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(64) void k_indep(float* out, int iters) {
  float a0 = threadIdx.x * 0.001f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 0.9999f, c = 1e-4f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      a0 = fmaf(a0, b, c); a1 = fmaf(a1, b, c); a2 = fmaf(a2, b, c); a3 = fmaf(a3, b, c);
      a4 = fmaf(a4, b, c); a5 = fmaf(a5, b, c); a6 = fmaf(a6, b, c); a7 = fmaf(a7, b, c);
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ __launch_bounds__(64) void k_chain(float* out, int iters) {
  float a = threadIdx.x * 0.001f;
  const float b = 0.9999f, c = 1e-4f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 64; u++) a = fmaf(a, b, c);
  }
  out[blockIdx.x * 64 + threadIdx.x] = a;
}
template <class K> static void run(const char* name, K k, int iters) {
  float* d; hipMalloc(&d, 8192 * 64 * 4); hipMemset(d, 0, 8192 * 64 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 8; wps++) {
    int wg = 1024 * wps;
    hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, iters); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) {
      hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double instr = (double)wg * iters * 64.0;
    printf("%-6s %d waves/SIMD (%5d WGs): %8.3f ms  %8.1f G wave-instr/s  (%.2f cycles per instruction and SIMD at 2.4 GHz)\n", name, wps, wg, best, instr / best / 1e6,
           1024.0 * 2.4e9 / (instr / (best * 1e-3)));
  }
  hipFree(d);
}
int main() { run("indep", k_indep, 20000); run("chain", k_chain, 20000); return 0; }
