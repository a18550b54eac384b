// How much VALU issue does a SIMD have beyond what ONE wavefront can use?  The same independent-FMA loop run by 1, 2, 3 and 4
// wavefronts per SIMD (blocks of 256 / 512 / 768 / 1024 threads on one CU): s_memtime cycles per instruction per wavefront.
// hipcc --offload-arch=gfx950 -O3 -o occupancy occupancy.hip && ./occupancy
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int KIND>
__global__ void k(long long* out, double* sink, double a0, double b0) {
  double a = a0 + threadIdx.x, b = b0, c = a0 * 2, d = a0 * 3, e = a0 * 5;
  float fa = (float)a0, fc = 2.f, fd = 3.f, fe = 5.f, fb = (float)b0;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 64; it++) {
    if (KIND == 0) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
    if (KIND == 1) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(fa), "+v"(fc), "+v"(fd), "+v"(fe) : "v"(fb));) }
    if (KIND == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  sink[threadIdx.x] = a + c + d + e + fa + fc + fd + fe;
}
int main() {
  long long* out; double* sink;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&sink, 2048 * 8);
  const char* nm[] = {"4 independent v_fma_f64 chains", "4 independent v_fma_f32 chains", "1 dependent v_fma_f64 chain"};
  for (int kind = 0; kind < 3; kind++)
    for (int waves = 4; waves <= 16; waves += 4) {
      for (int rep = 0; rep < 2; rep++) {
        if (kind == 0) k<0><<<1, 64 * waves>>>(out, sink, 1.0000001, 0.999999);
        if (kind == 1) k<1><<<1, 64 * waves>>>(out, sink, 1.0000001, 0.999999);
        if (kind == 2) k<2><<<1, 64 * waves>>>(out, sink, 1.0000001, 0.999999);
        (void)hipDeviceSynchronize();
      }
      long long h[16]; (void)hipMemcpy(h, out, 16 * 8, hipMemcpyDeviceToHost);
      double mx = 0; for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
      const double per = mx / (64.0 * 64 * 4);
      printf("%-34s %d wavefront(s) per SIMD: %.2f cycles per instruction per wavefront -> %.3f instructions / cycle / SIMD\n", nm[kind], waves / 4, per, (waves / 4) / per);
    }
  return 0;
}
