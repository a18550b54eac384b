// Does the operand kind of an fp64 FMA change its issue cost for a lone wavefront?  (VGPR pairs vs one SGPR pair / literal)
// hipcc --offload-arch=gfx950 -O3 -o operands operands.hip && ./operands
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
template <int KIND>
__global__ void k(long long* out, double* sink, double a0, double b0) {
  double a = a0 + threadIdx.x, b = b0, c = a0 * 2, d = a0 * 3, e = a0 * 5, f = b0 * 7, g = b0 * 11;
  long long t0 = clock64();
  for (int it = 0; it < 64; it++) {
    // four independent chains, per instruction
    if (KIND == 0) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(f));) }   // three VGPR pairs
    if (KIND == 1) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "s"(f));) }   // two VGPR pairs + an SGPR pair
    if (KIND == 2) { REP64(asm volatile("v_fma_f64 %0, %0, %4, 1.0\n v_fma_f64 %1, %1, %4, 1.0\n v_fma_f64 %2, %2, %4, 1.0\n v_fma_f64 %3, %3, %4, 1.0" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }            // inline constant
    if (KIND == 3) { REP64(asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b), "v"(f));) }                      // VOP2 form
    if (KIND == 4) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(f)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(f)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(f)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(f));) }  // dependent, three VGPR pairs
    if (KIND == 5) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(f)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(f)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(f)); asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "s"(f));) }  // dependent, SGPR addend
    if (KIND == 6) { REP64(asm volatile("s_mov_b32 s20, 0x12345678\n s_mov_b32 s21, 0x3ff12345\n v_fma_f64 %0, %0, %1, s[20:21]\n v_mul_f64 %2, %2, %1" : "+v"(a), "+v"(c) : "v"(b), "v"(f) : "s20", "s21");) }  // literal materialised every time (per 4 instructions)
  }
  long long t1 = clock64();
  if (threadIdx.x == 0) out[KIND] = t1 - t0;
  sink[threadIdx.x] = a + c + d + e + g;
}
int main() {
  long long* out; double* sink;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&sink, 1024 * 8);
  (void)hipMemset(out, 0, 64 * 8);
  for (int rep = 0; rep < 2; rep++) {
    k<0><<<1, 64>>>(out, sink, 1.0000001, 0.999999); k<1><<<1, 64>>>(out, sink, 1.0000001, 0.999999); k<2><<<1, 64>>>(out, sink, 1.0000001, 0.999999);
    k<3><<<1, 64>>>(out, sink, 1.0000001, 0.999999); k<4><<<1, 64>>>(out, sink, 1.0000001, 0.999999); k<5><<<1, 64>>>(out, sink, 1.0000001, 0.999999);
    k<6><<<1, 64>>>(out, sink, 1.0000001, 0.999999);
  }
  (void)hipDeviceSynchronize();
  long long h[7]; (void)hipMemcpy(h, out, 7 * 8, hipMemcpyDeviceToHost);
  const char* nm[] = {"indep fma, 3 VGPR pairs", "indep fma, 2 VGPR + SGPR pair", "indep fma, inline constant", "indep fmac (VOP2)", "dep fma, 3 VGPR pairs", "dep fma, SGPR addend", "2 s_mov + fma + mul (4 instr)"};
  for (int i = 0; i < 7; i++) printf("%-34s %.2f cycles per instruction\n", nm[i], h[i] / (64.0 * 256));
  return 0;
}
