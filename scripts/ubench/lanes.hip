// Does a wavefront with fewer active lanes issue fp64 VALU instructions faster?  (one wavefront, exec = the low N lanes)
// hipcc --offload-arch=gfx950 -O3 -o lanes lanes.hip && ./lanes
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))
template <int KIND>
__global__ void k(long long* out, double* sink, double a0, double b0) {
  double a = a0 + threadIdx.x, b = b0, c = a0 * 2, d = a0 * 3, e = a0 * 5;
  float fa = (float)a0, fb = (float)b0, fc = fa * 2, fd = fa * 3, fe = fa * 5;
  long long t0 = clock64();
  for (int it = 0; it < 16; it++) {
    if (KIND == 0) { REP256(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 1) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
    if (KIND == 2) { REP256(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa) : "v"(fb));) }
    if (KIND == 3) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(fa), "+v"(fc), "+v"(fd), "+v"(fe) : "v"(fb));) }
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[KIND] = t1 - t0;
  sink[threadIdx.x] = a + c + d + e + fa + fc + fd + fe;
}
int main() {
  long long* out; double* sink;
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1024 * 8);
  const char* nm[] = {"dep v_fma_f64", "indep v_fma_f64", "dep v_fma_f32", "indep v_fma_f32"};
  for (int lanes : {64, 32, 16, 8, 1}) {
    hipMemset(out, 0, 64 * 8);
    for (int rep = 0; rep < 2; rep++) { k<0><<<1, lanes>>>(out, sink, 1.0000001, 0.999999); k<1><<<1, lanes>>>(out, sink, 1.0000001, 0.999999); k<2><<<1, lanes>>>(out, sink, 1.0000001, 0.999999); k<3><<<1, lanes>>>(out, sink, 1.0000001, 0.999999); }
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, out, 4 * 8, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) printf("%2d lanes  %-18s %.2f cycles per instruction\n", lanes, nm[i], h[i] / (16.0 * 256));
  }
  return 0;
}
