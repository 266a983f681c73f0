// Single-wavefront LDS throughput: ns per batch of independent reads of 16 floats per lane (row-per-lane access), by instruction width and row stride.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/ldsbw.hip -o tools/ubench/ldsbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 2048
__global__ __launch_bounds__(64) void k(float* out, long long* t, int mode) {
  __shared__ __attribute__((aligned(16))) float fl[64 * 72 + 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 72 + 64; i += 64) fl[i] = 1.0f + i * 1e-6f;
  __syncthreads();
  float acc = 0.f;
  int p = 0;
  long long w0 = wall_clock64();
  for (int i = 0; i < N; i++) {
    float v[16];
    if (mode == 0) {          // 16 x ds_read_b32, stride 65
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = fl[lane * 65 + p + u];
    } else if (mode == 1) {   // 4 x ds_read_b128, stride 68 (16-byte aligned rows)
      const float4* q = (const float4*)(fl + lane * 68 + p * 4);
#pragma unroll
      for (int u = 0; u < 4; u++) { float4 x = q[u]; v[4 * u] = x.x; v[4 * u + 1] = x.y; v[4 * u + 2] = x.z; v[4 * u + 3] = x.w; }
    } else if (mode == 2) {   // 8 x ds_read_b64, stride 66
      const float2* q = (const float2*)(fl + lane * 66 + p * 2);
#pragma unroll
      for (int u = 0; u < 8; u++) { float2 x = q[u]; v[2 * u] = x.x; v[2 * u + 1] = x.y; }
    } else if (mode == 3) {   // 16 x ds_read_b32 broadcast (every lane the same address)
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = fl[p + u];
    } else if (mode == 4) {   // 16 x ds_read_b32, stride 64 (64-way... all lanes one bank)
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = fl[lane * 64 + p + u];
    } else if (mode == 5) {   // 16 x ds_read_b32, stride 1 (coalesced columns: lane = column)
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = fl[(p + u) * 65 + lane];
    } else if (mode == 6) {   // 16 x ds_write_b32 stride 65 then one read
#pragma unroll
      for (int u = 0; u < 16; u++) fl[lane * 65 + p + u] = acc + u;
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = 0.f;
      v[0] = fl[lane * 65 + p];
    } else if (mode == 7) {   // 4 x ds_write_b128 stride 68 then one read
      float4* q = (float4*)(fl + lane * 68 + p * 4);
#pragma unroll
      for (int u = 0; u < 4; u++) q[u] = make_float4(acc, acc + 1, acc + 2, acc + u);
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = 0.f;
      v[0] = fl[lane * 68 + p];
    } else if (mode == 8) {   // MFMA operand pattern: rows 4c + (lane >> 4), stride 65, col lane & 15: 16 reads
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = fl[(4 * u + (lane >> 4)) * 65 + (lane & 15) + p];
    } else if (mode == 9) {   // same with stride 68
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = fl[(4 * u + (lane >> 4)) * 68 + (lane & 15) + p];
    } else {                  // same with stride 80 (16 r + col: conflict-free for the four rows)
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = fl[((4 * u + (lane >> 4)) * 80 + (lane & 15) + p) % (64 * 72)];
    }
    float s0 = (v[0] + v[1]) + (v[2] + v[3]), s1 = (v[4] + v[5]) + (v[6] + v[7]), s2 = (v[8] + v[9]) + (v[10] + v[11]), s3 = (v[12] + v[13]) + (v[14] + v[15]);
    acc += (s0 + s1) + (s2 + s3);
    p = (int)acc & 1;   // next batch's addresses depend on this one
  }
  long long w1 = wall_clock64();
  out[blockIdx.x * 64 + lane] = acc + p;
  if (lane == 0 && blockIdx.x == 0) t[0] = w1 - w0;
}
int main() {
  float* d; long long* t; (void)hipMalloc(&d, 1024 * 64 * 4); (void)hipMalloc(&t, 16);
  const char* names[] = {"16 x ds_read_b32, row per lane, stride 65", "4 x ds_read_b128, row per lane, stride 68", "8 x ds_read_b64, row per lane, stride 66", "16 x ds_read_b32 broadcast",
                         "16 x ds_read_b32, stride 64 (all lanes one bank)", "16 x ds_read_b32, lane = column", "16 x ds_write_b32 stride 65 + 1 read", "4 x ds_write_b128 stride 68 + 1 read",
                         "16 x MFMA-operand read, stride 65", "16 x MFMA-operand read, stride 68", "16 x MFMA-operand read, stride 80"};
  for (int wg : {1, 512}) for (int mode = 0; mode <= 10; mode++) {
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k, dim3(wg), dim3(64), 0, 0, d, t, mode); (void)hipDeviceSynchronize(); }
    long long tt; (void)hipMemcpy(&tt, t, 8, hipMemcpyDeviceToHost);
    printf("WGs %3d  %-52s %7.1f ns per batch\n", wg, names[mode], tt * 10.0 / N);
  }
  return 0;
}
