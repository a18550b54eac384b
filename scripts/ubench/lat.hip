// Issue/latency micro-benchmark for the instruction kinds of the backward chain (one wavefront per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o lat lat.hip && ./lat
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))
#define REP256(x) REP4(REP64(x))

template <int KIND>
__global__ void k(long long* out, double* sink, double a0, double b0) {
  double a = a0 + threadIdx.x, b = b0, c = a0 * 2, d = a0 * 3, e = a0 * 5;
  float fa = (float)a0 + threadIdx.x, fb = (float)b0;
  int ia = threadIdx.x;
  long long t0 = __builtin_readcyclecounter();
  t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 16; it++) {
    if (KIND == 0) { REP256(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 1) { REP64(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));) }
    if (KIND == 2) { REP256(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(fa) : "v"(fb));) }
    if (KIND == 3) { REP256(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 4) { REP256(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 5) { REP256(asm volatile("s_nop 0\n v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(ia));) }
    if (KIND == 6) { REP256(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(a), "v"(b), "v"(ia), "v"(ia) : "vcc");) }
    if (KIND == 7) { REP256(asm volatile("v_rcp_f64 %0, %0" : "+v"(a));) }
    if (KIND == 8) { REP256(asm volatile("v_max_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
    if (KIND == 9) { REP64(asm volatile("v_fma_f64 %0, %0, %2, %2\n v_fma_f64 %1, %1, %2, %2\n" : "+v"(a), "+v"(c) : "v"(b));
                     asm volatile("v_fma_f64 %0, %0, %2, %2\n v_fma_f64 %1, %1, %2, %2\n" : "+v"(a), "+v"(c) : "v"(b));) }
    if (KIND == 10) { REP256(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(ia) : "v"(ia));) }
    if (KIND == 11) { REP256(asm volatile("s_add_u32 s20, s20, 1" : : : "s20", "scc");) }
    if (KIND == 12) { REP256(asm volatile("v_fma_f64 %0, %0, %1, %1\n s_add_u32 s20, s20, 1" : "+v"(a) : "v"(b) : "s20", "scc");) }
    if (KIND == 13) { REP256(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(fa): "v"(fb) : "vcc");) }  // dependent cmp -> select -> cmp
    if (KIND == 14) { REP256(asm volatile("v_fma_f32 %0, %0, %1, %1\n s_nop 1\n v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(fa) : "v"(fb));) }
    if (KIND == 15) { REP256(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "=v"(ia) : "v"(threadIdx.x));) }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[KIND] = t1 - t0;
  sink[threadIdx.x] = a + c + d + e + fa + ia;
}
int main() {
  long long* out; double* sink;
  hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1024 * 8);
  hipMemset(out, 0, 64 * 8);
#define RUN(K) k<K><<<1, 64>>>(out, sink, 1.0000001, 0.999999); k<K><<<1, 64>>>(out, sink, 1.0000001, 0.999999); { hipError_t e = hipDeviceSynchronize(); fprintf(stderr, "kind %d: %s\n", K, hipGetErrorString(e)); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15)
  long long h[64]; hipMemcpy(h, out, 64 * 8, hipMemcpyDeviceToHost);
  const char* nm[] = {"dep v_fma_f64", "4 indep v_fma_f64 chains (per instr)", "dep v_fma_f32", "dep v_mul_f64", "dep v_add_f64", "dep s_nop+dpp mov", "indep cmp_f64+cndmask (2 instr)",
                      "dep v_rcp_f64", "dep v_max_f64", "2 v_fma_f64 chains (per instr)", "dep bpermute+wait", "s_add", "fma_f64 + s_add (2 instr)",
                      "dep cmp+cndmask (2 instr)", "fma_f32 -> nop 1 -> dpp (3 instr)", "indep dpp mov"};
  for (int i = 0; i < 16; i++) printf("%-45s %.2f cycles (s_memtime) per statement\n", nm[i], h[i] / (16.0 * 256));
  int khz; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0); printf("clock %d kHz\n", khz);
  hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0); printf("wall clock %d kHz\n", khz);
  return 0;
}
