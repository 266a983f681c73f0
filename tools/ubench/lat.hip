// Latency calibration for the one-wavefront-per-SIMD regime of the fused kernel: ns per DEPENDENT operation of each kind (one workgroup of 64 lanes,
// nothing else on the CU to hide latency behind).  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/lat.hip -o tools/ubench/lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 4096
template <int K> __device__ __forceinline__ float rbcast(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + K, 0xf, 0xf, true));
}
__device__ __forceinline__ float rl(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k(float* out, long long* t, const int* gchain, int mode) {
  __shared__ int chain[4096];
  __shared__ float fl[4096 + 64 * 65];
  const int lane = threadIdx.x;
  for (int i = lane; i < 4096; i += 64) { chain[i] = (i * 67 + 13) & 4095; fl[i] = 1.0f + i * 1e-6f; }
  __syncthreads();
  float a = lane * 0.001f + 1.f, acc = 0.f;
  int p = lane;
  v4f m = {0.f, 0.f, 0.f, 0.f};
  long long w0 = wall_clock64();
  if (mode == 0) { for (int i = 0; i < N; i++) p = chain[p]; }                                                   // dependent ds_read_b32
  else if (mode == 1) { for (int i = 0; i < N; i++) a = __shfl(a, (lane + 1) & 63) + 1e-6f; }                    // dependent ds_bpermute
  else if (mode == 2) { for (int i = 0; i < N; i += 4) { a = fmaf(a, rl(a, 3), 1e-6f); a = fmaf(a, rl(a, 5), 1e-6f); a = fmaf(a, rl(a, 7), 1e-6f); a = fmaf(a, rl(a, 9), 1e-6f); } }   // readlane -> fma
  else if (mode == 3) { for (int i = 0; i < N; i += 4) { a = fmaf(a, rbcast<3>(a), 1e-6f); a = fmaf(a, rbcast<5>(a), 1e-6f); a = fmaf(a, rbcast<7>(a), 1e-6f); a = fmaf(a, rbcast<9>(a), 1e-6f); } }   // DPP fma
  else if (mode == 4) { for (int i = 0; i < N; i += 4) { a = fmaf(a, 0.999f, 1e-6f); a = fmaf(a, 0.999f, 1e-6f); a = fmaf(a, 0.999f, 1e-6f); a = fmaf(a, 0.999f, 1e-6f); } }   // plain fma
  else if (mode == 5) { for (int i = 0; i < N; i += 4) { m = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, m, 0, 0, 0); m = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, m, 0, 0, 0); m = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, m, 0, 0, 0); m = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, m, 0, 0, 0); } }
  else if (mode == 6) { for (int i = 0; i < N; i += 4) { a = __builtin_amdgcn_rsqf(a) + 1.f; a = __builtin_amdgcn_rsqf(a) + 1.f; a = __builtin_amdgcn_rsqf(a) + 1.f; a = __builtin_amdgcn_rsqf(a) + 1.f; } }   // rsq + add
  else if (mode == 7) { for (int i = 0; i < N; i++) p = gchain[p]; }                                              // dependent global load (L2 / L1 hit)
  else if (mode == 8) { for (int i = 0; i < N; i += 16) { float v[16];                                            // 16 independent LDS reads (stride-65 rows), then 16 dependent fmas
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = fl[4096 + lane * 65 + ((p + u) & 15)];
#pragma unroll
      for (int u = 0; u < 16; u++) acc = fmaf(v[u], 0.5f, acc);
      p = (int)acc & 15; } }
  else if (mode == 9) { for (int i = 0; i < N; i++) { fl[lane] = a; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); a = fl[(lane + 1) & 63] + 1e-6f; } }   // LDS write -> read round trip
  else if (mode == 10) { for (int i = 0; i < N; i += 4) { m = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, m, 0, 0, 0); a = m[0]; m = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, m, 0, 0, 0); a = m[1]; m = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, m, 0, 0, 0); a = m[2]; m = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, m, 0, 0, 0); a = m[3]; } }   // mfma -> VALU use -> mfma
  long long w1 = wall_clock64();
  out[blockIdx.x * 64 + lane] = a + acc + p + m[0] + m[1] + m[2] + m[3];
  if (lane == 0 && blockIdx.x == 0) t[0] = w1 - w0;
}
int main() {
  float* d; long long* t; int* g; hipMalloc(&d, 1024 * 64 * 4); hipMalloc(&t, 16); hipMalloc(&g, 4096 * 4);
  int h[4096]; for (int i = 0; i < 4096; i++) h[i] = (i * 67 + 13) & 4095; hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  const char* names[] = {"dependent ds_read_b32", "dependent ds_bpermute", "v_readlane -> v_fma", "DPP row_newbcast fma", "plain v_fma", "mfma 16x16x4 f32 (acc chain)", "v_rsq + v_add",
                         "dependent global_load (L1/L2 hit)", "batch of 16 LDS reads + 16 fmas (per batch)", "LDS write -> read round trip", "mfma -> VALU read -> mfma"};
  for (int wg : {1, 256}) for (int mode = 0; mode <= 10; mode++) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, t, g, mode); hipDeviceSynchronize(); }
    long long tt; hipMemcpy(&tt, t, 8, hipMemcpyDeviceToHost);
    const double n = mode == 8 ? N / 16 : N;
    printf("WGs %3d  %-48s %7.1f ns per op\n", wg, names[mode], tt * 10.0 / n);
  }
  return 0;
}
