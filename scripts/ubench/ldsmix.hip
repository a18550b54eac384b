// LDS cost of the hex chain's access pattern (scripts/ubench): per wavefront and step 13 record reads (8 ds_read_b128 + 5 ds_read_b64)
// from a ring, one exchange (2 ds_write_b64, 3 ds_read_b128), with 1 / 4 / 8 wavefronts per CU.
// hipcc --offload-arch=gfx950 -O3 -o ldsmix ldsmix.hip && ./ldsmix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(512) void k(long long* out, double* sink, int iters) {
  __shared__ double ring[4][24 * 192];
  __shared__ double xch[8][160];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lp = lane >> 4, a = (lane >> 2) & 3, b = lane & 3;
  for (int i = threadIdx.x; i < 4 * 24 * 192; i += blockDim.x) (&ring[0][0])[i] = 1.0 + i * 1e-6;
  for (int i = threadIdx.x; i < 8 * 160; i += blockDim.x) (&xch[0][0])[i] = 0.5;
  __syncthreads();
  double acc = lane, i0 = lane, i1 = lane + 1, i2 = lane + 2, i3 = lane + 3;
  const double* r0 = &ring[wave & 3][0] + lp * 2;
  double* xw = &xch[wave][16 * lp];
  long long t0 = __builtin_amdgcn_s_memtime();
  int slot = 0;
  for (int it = 0; it < iters; it++) {
    const double* r = r0 + slot * 192;
    slot = (slot + 1 == 24) ? 0 : slot + 1;
    if (MODE & 1) {  // the record reads
      d2 p0 = *(const d2*)(r + (2 * a) * 8), p1 = *(const d2*)(r + (2 * a + 1) * 8), p2 = *(const d2*)(r + (2 * b) * 8), p3 = *(const d2*)(r + (2 * b + 1) * 8);
      d2 p4 = *(const d2*)(r + 8 * 8), p5 = *(const d2*)(r + 9 * 8), p6 = *(const d2*)(r + 22 * 8), p7 = *(const d2*)(r + 23 * 8);
      double s0 = r[(10) * 8 + (b & 1) + (b >> 1) * 8], s1 = r[(12 + ((a + 4 * b) >> 1)) * 8 + (a & 1)], s2 = r[(12 + ((b + 4 * a) >> 1)) * 8 + (b & 1)];
      double s3 = r[(20 + (a >> 1)) * 8 + (a & 1)], s4 = r[(20 + (b >> 1)) * 8 + (b & 1)];
      acc += p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + s0 + s1 + s2 + s3 + s4;
    }
    if (MODE & 2) {  // one exchange round trip
      xw[4 * b + a] = acc;
      xw[64 - 16 * lp + 4 * lp + a] = acc * 0.5;
      __builtin_amdgcn_sched_barrier(0);
      d4 c0 = *(const d4*)(xw + 4 * b), c1 = *(const d4*)(xw + 4 * a), c2 = *(const d4*)(xw + 64 - 16 * lp + 4 * lp);
      acc = acc * 0.999 + c0.x + c1.y + c2.z;
    }
    if (MODE & 4) {  // a second dependent round trip
      xw[4 * a + b] = acc;
      __builtin_amdgcn_sched_barrier(0);
      d4 c0 = *(const d4*)(xw + 4 * a);
      acc = acc * 0.999 + c0.w;
    }
    if (MODE & 16) {  // 100 FMAs in four chains that do NOT depend on the loads: does LDS issue overlap VALU issue?
#pragma unroll
      for (int q = 0; q < 25; q++) {
        i0 = __builtin_fma(i0, 0.9999, 1e-9);
        i1 = __builtin_fma(i1, 0.9999, 1e-9);
        i2 = __builtin_fma(i2, 0.9999, 1e-9);
        i3 = __builtin_fma(i3, 0.9999, 1e-9);
      }
    }
    if (MODE & 8) {  // ~200 dependent-free VALU instructions of other work
#pragma unroll
      for (int q = 0; q < 50; q++) acc = __builtin_fma(acc, 0.9999, 1e-9);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + i0 + i1 + i2 + i3;
}
template <int MODE>
void run(const char* what, int waves) {
  long long* out; double* sink;
  hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 256 * 512 * 8);
  const int iters = 2000;
  k<MODE><<<256, 64 * waves>>>(out, sink, iters);
  k<MODE><<<256, 64 * waves>>>(out, sink, iters);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, out, 8 * 8, hipMemcpyDeviceToHost);
  printf("%-60s %d waves/CU: %.0f cycles per step (wave 0)\n", what, waves, (double)h[0] / iters);
  hipFree(out); hipFree(sink);
}
int main() {
  for (int w : {1, 4, 8}) {
    if (w == 1) { run<16>("100 independent fma", 1); run<17>("13 record reads + 100 independent fma", 1); run<1>("13 record reads", 1); run<2>("one exchange (2 writes, 3 b128 reads)", 1); run<3>("reads + exchange", 1); run<7>("reads + two exchanges", 1); run<15>("reads + two exchanges + 50 fma", 1); run<8>("50 dependent fma only", 1); }
    if (w == 4) { run<17>("13 record reads + 100 independent fma", 4); run<1>("13 record reads", 4); run<2>("one exchange (2 writes, 3 b128 reads)", 4); run<3>("reads + exchange", 4); run<7>("reads + two exchanges", 4); run<15>("reads + two exchanges + 50 fma", 4); }
    if (w == 8) { run<1>("13 record reads", 8); run<2>("one exchange (2 writes, 3 b128 reads)", 8); run<3>("reads + exchange", 8); run<7>("reads + two exchanges", 8); run<15>("reads + two exchanges + 50 fma", 8); }
  }
  return 0;
}
