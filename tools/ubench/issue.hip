// microbenchmark: how does a VALU-only / readlane-heavy / LDS-heavy wave scale when 1..8 single-wave workgroups share a CU?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(64) void k_valu(float* out, int iters) {
  float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { a = fmaf(a, b, c); d = fmaf(d, b, a); c = fmaf(c, b, d); b = fmaf(b, 0.99999f, 1e-7f); }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(64) void k_readlane(float* out, int iters) {
  float a = threadIdx.x * 0.001f, acc = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) { float s = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), u)); acc = fmaf(a, s, acc); a = fmaf(a, 0.9999f, 1e-6f); }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_lds(float* out, int iters) {
  __shared__ float sm[64 * 17];
  float acc = 0.f;
  for (int i = 0; i < 17; i++) sm[threadIdx.x * 17 + i] = threadIdx.x + i;
  __syncthreads();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) acc = fmaf(sm[threadIdx.x * 17 + u], 1.0001f, acc);
    sm[threadIdx.x * 17 + (i & 15)] = acc;
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_trans(float* out, int iters) {
  float a = threadIdx.x * 0.001f + 1.f, acc = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) { acc += __builtin_amdgcn_sqrtf(a) + __builtin_amdgcn_rcpf(a + 1.f); a = fmaf(a, 1.0001f, 1e-3f); }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_bperm(float* out, int iters) {
  float a = threadIdx.x * 0.001f, acc = 0.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) { acc += __shfl(a, (threadIdx.x + u + i) & 63); a = fmaf(a, 0.9999f, 1e-6f); }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
__global__ __launch_bounds__(64) void k_gload(float* out, int iters) {
  const float* tab = out + 8192 * 64;   // 32 KB table, L1/L2 resident
  float acc = 0.f; int idx = threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) { acc += tab[(idx + 64 * u) & 8191]; }
    idx = (idx + 17) & 8191;
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_mfma(float* out, int iters) {
  v4f acc = {0, 0, 0, 0}; float a = threadIdx.x * 0.001f, b = 1.f;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++) { acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0); a = fmaf(a, 0.999f, 1e-4f); }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}
__global__ __launch_bounds__(64) void k_ldsrw(float* out, int iters) {
  __shared__ float sm[64 * 17];
  float acc = threadIdx.x;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) { sm[threadIdx.x * 17 + u] = acc; __syncthreads(); acc = fmaf(sm[((threadIdx.x + 1) & 63) * 17 + u], 0.999f, 1.f); __syncthreads(); }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc;
}
template <class K> static void run(const char* name, K k, int iters) {
  float* d; hipMalloc(&d, 8192 * 64 * 4 + 32768); hipMemset(d, 0, 8192 * 64 * 4 + 32768);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg = 256; wg <= 2048; wg *= 2) {
    hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, iters); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s WGs %5d (%.0f per CU): %8.3f ms\n", name, wg, wg / 256.0, ms);
  }
  hipFree(d);
}
int main() { run("valu", k_valu, 20000); run("trans", k_trans, 20000); run("bperm", k_bperm, 20000); run("gload", k_gload, 20000); run("mfma", k_mfma, 20000); run("ldsrw", k_ldsrw, 5000); run("lds", k_lds, 20000); return 0; }
