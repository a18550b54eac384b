// v_mfma_f64_4x4x4_4b_f64 on gfx950: operand / result lane layouts, summation order, issue and dependent latency.
// hipcc --offload-arch=gfx950 -O3 -o mfma4 mfma4.hip && ./mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k_layout(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}
template <int DEP>
__global__ void k_time(double* out, long long* cyc, double s) {
  double a = 1.0 + threadIdx.x * 1e-3, b = 0.5, c0 = s, c1 = s + 1, c2 = s + 2, c3 = s + 3;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < 256; it++) {
    if (DEP) {  // D feeds the next B operand (the chain of the Riccati step: W -> Qxx -> ...)
#pragma unroll
      for (int q = 0; q < 8; q++) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, c0, 0.0, 0, 0, 0);
    } else {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = c0 + c1 + c2 + c3;
  if (threadIdx.x == 0) cyc[DEP] = t1 - t0;
}
int main() {
  double ha[64], hb[64], hc[64], hd[64], *a, *b, *c, *d;
  long long* cyc;
  hipMalloc(&a, 512); hipMalloc(&b, 512); hipMalloc(&c, 512); hipMalloc(&d, 512); hipMalloc(&cyc, 64);
  // 1. which lane supplies A[blk][i][k] / B[blk][k][j], which lane receives D[blk][i][j]: probe with unit impulses
  //    A = delta at lane la, B = delta at lane lb -> the lanes of D that light up
  int Ai[64], Ak[64], Ab[64], Bk[64], Bj[64], Bb[64];
  for (int l = 0; l < 64; l++) Ai[l] = Ak[l] = Ab[l] = Bk[l] = Bj[l] = Bb[l] = -1;
  int hits[64][64];
  for (int la = 0; la < 64; la++)
    for (int lb = 0; lb < 64; lb++) {
      for (int l = 0; l < 64; l++) { ha[l] = (l == la); hb[l] = (l == lb); hc[l] = 0; }
      hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice); hipMemcpy(c, hc, 512, hipMemcpyHostToDevice);
      k_layout<<<1, 64>>>(a, b, c, d);
      hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
      hits[la][lb] = -1;
      for (int l = 0; l < 64; l++) if (hd[l] != 0) hits[la][lb] = l;
    }
  printf("D lane lit by (A impulse at lane la, B impulse at lane lb); -1 = none.  Rows la = 0..63, only lb with a hit listed:\n");
  for (int la = 0; la < 64; la++) {
    printf("la %2d:", la);
    for (int lb = 0; lb < 64; lb++) if (hits[la][lb] >= 0) printf(" (lb %2d -> D %2d)", lb, hits[la][lb]);
    printf("\n");
  }
  // 2. summation order: D = C + sum_k a_k b_k with values whose sum depends on the order
  for (int l = 0; l < 64; l++) { ha[l] = 1.0; hb[l] = 0; hc[l] = 0; }
  // block 0, i = 0, j = 0: need B lanes of (k = 0..3, j = 0) -- filled after the layout is known; here: all B = terms by k guess lane>>4
  const double terms[4] = {1e16, 1.0, -1e16, 1.0};
  for (int l = 0; l < 64; l++) hb[l] = terms[l >> 4];
  hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice); hipMemcpy(c, hc, 512, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(a, b, c, d);
  hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
  printf("order probe (terms by lane>>4 = 1e16, 1, -1e16, 1): D[0] = %.17g   [k ascending from C: ((1e16+1)-1e16)+1 = 1 ; pairwise would differ]\n", hd[0]);
  const double t2[4] = {1.0, 1e16, 1.0, -1e16};
  for (int l = 0; l < 64; l++) hb[l] = t2[l >> 4];
  hipMemcpy(b, hb, 512, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(a, b, c, d);
  hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
  printf("order probe (terms 1, 1e16, 1, -1e16): D[0] = %.17g   [k ascending: ((1+1e16)+1)-1e16 = 0]\n", hd[0]);
  // fused or not: a*b exact product kept?  (1+2^-30)^2 - 1 - 2^-29 = 2^-60 only if the product is not rounded before the add
  for (int l = 0; l < 64; l++) { ha[l] = (l >> 4) == 0 ? 1.0 + ldexp(1.0, -30) : 0.0; hb[l] = (l >> 4) == 0 ? 1.0 + ldexp(1.0, -30) : 0.0; hc[l] = -(1.0 + ldexp(1.0, -29)); }
  hipMemcpy(a, ha, 512, hipMemcpyHostToDevice); hipMemcpy(b, hb, 512, hipMemcpyHostToDevice); hipMemcpy(c, hc, 512, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(a, b, c, d);
  hipMemcpy(hd, d, 512, hipMemcpyDeviceToHost);
  printf("fma probe: D[0] = %.17g (2^-60 = %.17g if each term is a fused multiply-add)\n", hd[0], ldexp(1.0, -60));
  // 3. timing
  k_time<0><<<1, 64>>>(d, cyc, 1.0); k_time<0><<<1, 64>>>(d, cyc, 1.0);
  k_time<1><<<1, 64>>>(d, cyc, 1.0); k_time<1><<<1, 64>>>(d, cyc, 1.0);
  long long hcy[2]; hipDeviceSynchronize(); hipMemcpy(hcy, cyc, 16, hipMemcpyDeviceToHost);
  printf("independent accumulators: %.1f cycles per mfma;  D -> B dependent chain: %.1f cycles per mfma\n", hcy[0] / (256.0 * 8), hcy[1] / (256.0 * 8));
  return 0;
}
