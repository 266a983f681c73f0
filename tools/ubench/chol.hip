// calibrate: register Cholesky (DPP row broadcast) factor + solve, cycles per call; s_memtime vs s_memrealtime tick rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define FMIN 1e-20f
template <int K> __device__ __forceinline__ float rbcast(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + K, 0xf, 0xf, true));
}
template <int N, int J, int K> struct Upd { static __device__ __forceinline__ void run(float (&a)[N], float lij) { if constexpr (K < N) { a[K] = fmaf(-lij, rbcast<K>(lij), a[K]); Upd<N, J, K + 1>::run(a, lij); } } };
template <int N, int J> struct Step { static __device__ __forceinline__ void run(float (&a)[N], float (&inv)[N]) { if constexpr (J < N) {
  const float iv = rsqrtf(fmaxf(rbcast<J>(a[J]), FMIN)); inv[J] = iv; const float lij = a[J] * iv; a[J] = lij; Upd<N, J, J + 1>::run(a, lij); Step<N, J + 1>::run(a, inv); } } };
template <int N, int K> struct Fwd { static __device__ __forceinline__ float run(const float (&a)[N], const float (&inv)[N], float x, int row) { if constexpr (K < N) {
  const float xk = rbcast<K>(x) * inv[K]; x = row == K ? xk : (row > K ? fmaf(-a[K], xk, x) : x); return Fwd<N, K + 1>::run(a, inv, x, row); } else return x; } };
__global__ __launch_bounds__(64) void k(float* out, long long* t, int iters) {
  const int lane = threadIdx.x, r = lane & 15;
  float acc = 0.f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    float a[16], inv[16];
#pragma unroll
    for (int k2 = 0; k2 < 16; k2++) a[k2] = (r == k2 ? 4.f + it * 1e-6f : 0.1f / (1 + r + k2)) + acc * 1e-9f;
    Step<16, 0>::run(a, inv);
    float x = Fwd<16, 0>::run(a, inv, (float)r, r);
    acc += x;
  }
  long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
int main() {
  float* d; long long* t; hipMalloc(&d, 4096 * 64 * 4); hipMalloc(&t, 16);
  for (int wg : {1, 1024}) {
    hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, t, 2000); hipDeviceSynchronize();
    hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, t, 2000); hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("WGs %d: factor+fwd  %.0f s_memtime ticks per call, %.2f us per call (realtime), ticks/us = %.0f\n", wg, h[0] / 2000.0, h[1] / 100.0 / 2000.0, h[0] / (h[1] / 100.0));
  }
  return 0;
}
