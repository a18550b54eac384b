// Do v_mfma_f64_16x16x4_f64 and fp64 VALU instructions of DIFFERENT wavefronts on one SIMD overlap, or do they share one
// datapath?  One block of 8 wavefronts on one CU (two per SIMD; role by the SIMD-local slot, read from HW_ID):
//   kind 0: every wavefront issues N dependent-free MFMAs (4 accumulator chains)          -> cycles per MFMA per SIMD
//   kind 1: every wavefront issues N independent v_fma_f64 (4 chains)                     -> cycles per FMA per SIMD
//   kind 2: on every SIMD one wavefront runs the MFMA loop, the other the FMA loop        -> max of the two = overlap, sum = none
//   kind 3: ONE wavefront per SIMD alternates 1 MFMA with F independent FMAs in one instruction stream
// hipcc --offload-arch=gfx950 -O3 -o coissue coissue.hip && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
__device__ __forceinline__ unsigned simd_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 4, 2)" : "=s"(v));  // SIMD_ID: bits 5:4
  return v;
}
__device__ __forceinline__ void mfma_loop(double4_t* acc, double a, double b, int iters) {
  for (int it = 0; it < iters; it++) {
    REP16(acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0); acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[2], 0, 0, 0); acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[3], 0, 0, 0);)
  }
}
__device__ __forceinline__ void fma_loop(double& a, double& c, double& d, double& e, double b, int iters) {
  for (int it = 0; it < iters; it++) {
    REP16(asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
  }
}
// iters_m MFMA rounds of 64, iters_f FMA rounds of 64
__global__ __launch_bounds__(512) void k(int kind, int iters_m, int iters_f, long long* out, double* sink, double a0, double b0) {
  __shared__ int slot[4];
  if (threadIdx.x < 4) slot[threadIdx.x] = 0;
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned sid = simd_id();
  int my = 0;
  if (lane == 0) my = atomicAdd(&slot[sid], 1);
  my = __shfl(my, 0, 64);  // 0 / 1: first / second wavefront that reported from this SIMD
  double a = a0 + lane, b = b0, c = a0 * 2, d = a0 * 3, e = a0 * 5;
  double4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  int role = -1;
  if (kind == 0) role = 0;
  if (kind == 1) role = 1;
  if (kind == 2) role = my;       // slot 0: MFMA, slot 1: FMA
  if (kind == 3) role = (my == 0) ? 2 : -1;
  if (role == 0) mfma_loop(acc, a, b, iters_m);
  if (role == 1) fma_loop(a, c, d, e, b, iters_f);
  if (role == 2) {
    for (int it = 0; it < iters_m; it++) {
      REP16(acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[0], 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[1], 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[2], 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b, acc[3], 0, 0, 0);
            asm volatile("v_fma_f64 %0, %0, %4, %4\n v_fma_f64 %1, %1, %4, %4\n v_fma_f64 %2, %2, %4, %4\n v_fma_f64 %3, %3, %4, %4" : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));)
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    out[3 * wave] = t1 - t0;
    out[3 * wave + 1] = role;
    out[3 * wave + 2] = sid;
  }
  double s = a + c + d + e;
  for (int q = 0; q < 4; q++) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  sink[threadIdx.x] = s;
}
int main() {
  long long* out; double* sink;
  (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&sink, 2048 * 8);
  const int IM = 16, IF = 16 * 12;  // 1024 MFMAs (64 cycles each if the pipe is 32 flop/cycle/SIMD) against 12288 FMAs
  for (int kind = 0; kind < 4; kind++) {
    for (int rep = 0; rep < 2; rep++) {
      k<<<1, 512>>>(kind, IM, kind == 3 ? IM : IF, out, sink, 1.0000001, 0.999999);
      (void)hipDeviceSynchronize();
    }
    long long h[24]; (void)hipMemcpy(h, out, 24 * 8, hipMemcpyDeviceToHost);
    printf("kind %d:", kind);
    for (int w = 0; w < 8; w++) printf("  [w%d simd%lld role%lld %lld cyc]", w, h[3 * w + 2], h[3 * w + 1], h[3 * w]);
    printf("\n");
    double tm = 0, tf = 0;
    for (int w = 0; w < 8; w++) { if (h[3 * w + 1] == 0 || h[3 * w + 1] == 2) tm = h[3 * w] > tm ? h[3 * w] : tm; if (h[3 * w + 1] == 1) tf = h[3 * w] > tf ? h[3 * w] : tf; }
    if (kind == 0) printf("  two MFMA wavefronts per SIMD: %.1f cycles per MFMA per SIMD (issue)\n", tm / (2.0 * IM * 64));
    if (kind == 1) printf("  two FMA wavefronts per SIMD: %.2f cycles per v_fma_f64 per SIMD\n", tf / (2.0 * IF * 64));
    if (kind == 2) printf("  one MFMA + one FMA wavefront per SIMD: MFMA loop %.0f cycles (%.1f per MFMA), FMA loop %.0f cycles (%.2f per FMA)\n", tm, tm / (IM * 64.0), tf, tf / (IF * 64.0));
    if (kind == 3) printf("  one wavefront per SIMD, 1 MFMA : 4 FMA interleaved: %.1f cycles per (MFMA + 4 FMA)\n", tm / (IM * 64.0));
  }
  return 0;
}
