// Micro-benchmarks of single-wave fp64 instruction latencies on gfx950 (dev tool, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define N 2048
template <int MODE>
__global__ void k(double* out, long long* cyc, double a, double b) {
  double x = a + threadIdx.x * 1e-9, y = b, z = a * 0.5, w = b * 0.25;
  double x2 = x + 1, x3 = x + 2, x4 = x + 3;
  long long t0 = __builtin_readcyclecounter();
  long long c0 = clock64();
#pragma unroll 1
  for (int i = 0; i < N / 8; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (MODE == 0) x = __builtin_fma(x, y, z);                       // dependent fma
      if (MODE == 1) { x = __builtin_fma(x, y, z); x2 = __builtin_fma(x2, y, z); x3 = __builtin_fma(x3, y, z); x4 = __builtin_fma(x4, y, z); }  // 4 indep chains
      if (MODE == 2) x = y / (x + w);                                  // dependent IEEE divide
      if (MODE == 3) x = sqrt(x * x + w);                              // dependent sqrt
      if (MODE == 4) { int lo = __double2loint(x), hi = __double2hiint(x); lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xf, 0xf, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xf, 0xf, true); x = __hiloint2double(hi, lo) + z; }  // dpp bcast + add
      if (MODE == 5) x = (x < y) ? x + z : x - w;                       // cmp+cndmask chain
      if (MODE == 6) x = __builtin_amdgcn_rcp(x + w);                   // dependent rcp
      if (MODE == 7) x = x + y;                                        // dependent add
      if (MODE == 8) x = fmax(fmin(x + z, y), w);                      // clamp chain
    }
  }
  long long c1 = clock64();
  long long t1 = __builtin_readcyclecounter();
  out[threadIdx.x + blockIdx.x * blockDim.x] = x + x2 + x3 + x4;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = c1 - c0; cyc[1] = t1 - t0; }
}
template <int MODE> void run(const char* name, int per) {
  double* out; long long* cyc; hipMalloc(&out, 64 * 8 * 1024); hipMalloc(&cyc, 16);
  k<MODE><<<1, 64>>>(out, cyc, 1.0000001, 0.9999999); hipDeviceSynchronize();
  k<MODE><<<1, 64>>>(out, cyc, 1.0000001, 0.9999999); hipDeviceSynchronize();
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-28s clock64 %.2f /op   s_memtime-ish %.2f /op\n", name, (double)h[0] / (N * per), (double)h[1] / (N * per));
  hipFree(out); hipFree(cyc);
}
int main() {
  run<0>("dependent fma_f64", 1); run<1>("4 independent fma_f64", 4); run<7>("dependent add_f64", 1);
  run<2>("dependent div (IEEE)+add", 1); run<3>("dependent sqrt+fma", 1); run<6>("dependent rcp+add", 1);
  run<4>("dpp bcast(2 mov)+add", 1); run<5>("cmp+cndmask+add chain", 1); run<8>("clamp(min,max)+add", 1);
  return 0;
}
