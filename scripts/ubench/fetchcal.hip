// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS path's access widths (MI355X_MICROARCH.md, HBM section:
// FETCH_SIZE is known to report half the bytes of a 16 B/lane coalesced stream; other widths are uncalibrated).
// Every kernel moves a known number of bytes once, over a buffer larger than the 256 MiB Infinity Cache:
//   k_read16   16 B per lane, fully coalesced (the guide's calibrated case)
//   k_read8row  8 B per lane, a wavefront reads four 128-byte ROWS of 16 trajectories at four distant addresses -- what every
//               load of the tiled [tile][t][e][16] layout looks like (chain, producers, rollouts)
//   k_read8     8 B per lane, 512 contiguous bytes per wavefront
//   k_write8row / k_write16   the same shapes as stores
// hipcc --offload-arch=gfx950 -O3 -o fetchcal fetchcal.hip ; rocprofv3 --pmc FETCH_SIZE --kernel-trace -- ./fetchcal   (and WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));
constexpr size_t kBytes = (size_t)1 << 30;  // 1 GiB per kernel

__global__ void k_read16(const d2* __restrict__ p, double* out, size_t n2) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    const d2 v = p[i];
    acc += v.x + v.y;
  }
  if (acc == 12345.678) out[0] = acc;
}
__global__ void k_read8(const double* __restrict__ p, double* out, size_t n) {
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 12345.678) out[0] = acc;
}
// rows of 16 doubles; the four 16-lane groups of a wavefront take rows a quarter of the buffer apart
__global__ void k_read8row(const double* __restrict__ p, double* out, size_t nrows) {
  const int lane = threadIdx.x & 63, l = lane & 15, g = lane >> 4;
  const size_t quarter = nrows / 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  double acc = 0;
  for (size_t r = wave; r < quarter; r += nwaves) acc += p[(g * quarter + r) * 16 + l];
  if (acc == 12345.678) out[0] = acc;
}
__global__ void k_write8row(double* __restrict__ p, size_t nrows) {
  const int lane = threadIdx.x & 63, l = lane & 15, g = lane >> 4;
  const size_t quarter = nrows / 4;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t r = wave; r < quarter; r += nwaves) p[(g * quarter + r) * 16 + l] = (double)r;
}
__global__ void k_write16(d2* __restrict__ p, size_t n2) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    d2 v;
    v.x = (double)i;
    v.y = 1.0;
    p[i] = v;
  }
}
int main() {
  double *buf, *out;
  if (hipMalloc(&buf, kBytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  hipMemset(buf, 0, kBytes);
  const size_t n = kBytes / 8;
  k_read16<<<2048, 256>>>((const d2*)buf, out, n / 2);
  k_read8<<<2048, 256>>>(buf, out, n);
  k_read8row<<<2048, 256>>>(buf, out, n / 16);
  k_write8row<<<2048, 256>>>(buf, n / 16);
  k_write16<<<2048, 256>>>((d2*)buf, n / 2);
  hipError_t e = hipDeviceSynchronize();
  printf("fetchcal: every kernel moved %zu bytes (%s)\n", kBytes, hipGetErrorString(e));
  return 0;
}
