// What does a vector memory instruction cost in the CU's address / L1 path?  (the rollouts issue ~11 per step and wavefront)
// hipcc --offload-arch=gfx950 -O3 -o vmem vmem.hip && ./vmem
// Every wavefront streams through a small L1/L2-resident buffer with independent loads of one shape; blocks = #CU,
// waves per CU varied; reported: cycles per load instruction and CU, and bytes per cycle and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
// shape 0: dwordx2, 64 lanes, 4 sub-groups of 16 lanes read the SAME 128 B row (the rollouts' pattern)
// shape 1: dwordx2, 64 lanes, 512 B distinct (fully coalesced)
// shape 2: dwordx4, 64 lanes, sub-groups read the same 256 B
// shape 3: dwordx4, 64 lanes, 1 KB distinct
// shape 4 / 5: dwordx2, only lanes 0..15 / 0..31 active
template <int SHAPE>
__global__ __launch_bounds__(512) void k(const double* __restrict__ buf, double* out, int iters, int rows, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l = lane & 15;
  double acc = 0;
  const double* base = buf + (size_t)blockIdx.x * rows * 128 + (wave & 0) ;  // one 16 KB (rows = 16) or 1 MB (rows = 1024) buffer per CU
  __syncthreads();
  const long long t0 = clock64();
  if (SHAPE < 4 || lane < (SHAPE == 4 ? 16 : 32))  // (shapes 4, 5: the whole loop under a 16- / 32-lane exec mask)
  for (int it = 0; it < iters; it++) {
    const double* p = base + (size_t)(it % (rows / 16)) * 16 * 128;
    double a[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      if (SHAPE == 0 || SHAPE >= 4) a[r] = p[r * 128 + l];
      if (SHAPE == 1) a[r] = p[r * 128 + lane];
      if (SHAPE == 2) { d2 v = *(const d2*)(p + r * 128 + 2 * l); a[r] = v.x + v.y; }
      if (SHAPE == 3) { d2 v = *(const d2*)(p + r * 128 + 2 * lane); a[r] = v.x + v.y; }
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) acc += a[r] * a[r + 1];
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int SHAPE>
void run(const char* name, int waves, const double* buf, double* out, long long* cyc, int bytes_per_instr, int rows) {
  const int blocks = 256, iters = 2000;
  k<SHAPE><<<blocks, 64 * waves>>>(buf, out, 10, rows, cyc);
  k<SHAPE><<<blocks, 64 * waves>>>(buf, out, iters, rows, cyc);
  hipDeviceSynchronize();
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; i++) s += h[i]; s /= blocks;
  const double instr = (double)iters * 16 * waves;
  printf("%-36s %4d KB/CU, waves/CU %d: %6.1f cycles per load instruction and CU, %6.1f B/cycle/CU (footprint)\n", name, rows, waves, s / instr, bytes_per_instr * instr / s);
}
int main() {
  double *buf, *out; long long* cyc;
  const size_t n = (size_t)256 * 1024 * 128;
  hipMalloc(&buf, n * 8); hipMemset(buf, 0, n * 8); hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8);
  for (int rows : {16, 1024})  // L1-resident; streaming from L2 / HBM (256 MB in all)
  for (int w : {1, 4, 8}) {
    run<0>("dwordx2, 4 x the same 128 B row", w, buf, out, cyc, 128, rows);
    run<1>("dwordx2, 512 B distinct", w, buf, out, cyc, 512, rows);
    run<2>("dwordx4, 4 x the same 256 B", w, buf, out, cyc, 256, rows);
    run<3>("dwordx4, 1 KB distinct", w, buf, out, cyc, 1024, rows);
    run<4>("dwordx2, 16 lanes active (128 B)", w, buf, out, cyc, 128, rows);
    run<5>("dwordx2, 32 lanes active (128 B x 2)", w, buf, out, cyc, 128, rows);
  }
  return 0;
}
