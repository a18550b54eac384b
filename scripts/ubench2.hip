// shader clock under a latency-bound, sparse (256-wave) fp64 load (dev tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(double* out, long long* cyc, double a, double b, int n) {
  double x = a + threadIdx.x * 1e-9, y = b, z = a * 0.5;
  long long c0 = clock64();
  long long w0 = wall_clock64();
#pragma unroll 1
  for (int i = 0; i < n; i++) {
#pragma unroll
    for (int j = 0; j < 16; j++) x = __builtin_fma(x, y, z);
  }
  long long c1 = clock64();
  long long w1 = wall_clock64();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = c1 - c0; cyc[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
  double* out; long long* cyc; hipMalloc(&out, 8 * 64 * 4096); hipMalloc(&cyc, 16 * 4096);
  int wc_khz = 0; hipDeviceGetAttribute(&wc_khz, hipDeviceAttributeWallClockRate, 0);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("wall clock rate %d kHz, clock rate attr %d kHz\n", wc_khz, clk_khz);
  for (int blocks : {1, 256, 1024, 4096}) {
    for (int rep = 0; rep < 2; rep++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      int n = 20000;
      hipEventRecord(e0); k<<<blocks, 64>>>(out, cyc, 1.0000001, 0.9999999, n); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
      printf("blocks %4d: %.3f ms, clock64 ticks %lld (%.2f/op), wall ticks %lld -> shader clock %.0f MHz (by event %.0f MHz)\n", blocks, ms, h[0],
             (double)h[0] / (n * 16.0), h[1], (double)h[0] / ((double)h[1] / wc_khz * 1e3) / 1e6 * 1e0, (double)h[0] / (ms * 1e-3) / 1e6);
    }
  }
  return 0;
}
