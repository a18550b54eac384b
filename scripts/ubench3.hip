// VMEM issue-cost model on gfx950: cycles per global load/store instruction for one wave, as a
// function of width, active lanes and address replication (dev tool, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double2_t __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(const double* __restrict__ in, double* __restrict__ out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  const size_t wbase = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (1 << 16);
  double acc = 0;
  long long t_issue = 0, t_total = 0;
  for (int it = 0; it < iters; it++) {
    const double* p = in + wbase + (size_t)it * 4096;
    double v[16];
    double2_t w[8];
    long long c0 = clock64();
    if (MODE == 0) {  // 16 x dwordx2, 64 distinct consecutive doubles
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = p[j * 64 + lane];
    } else if (MODE == 1) {  // 16 x dwordx2, addresses replicated 4x (16 distinct)
#pragma unroll
      for (int j = 0; j < 16; j++) v[j] = p[j * 64 + (lane & 15)];
    } else if (MODE == 2) {  // 8 x dwordx4, distinct
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = *(const double2_t*)(p + j * 128 + lane * 2);
    } else if (MODE == 3) {  // 16 x dwordx2 but only 16 lanes active
      if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = p[j * 64 + lane];
      }
    } else if (MODE == 4) {  // 8 x dwordx4 replicated 4x
#pragma unroll
      for (int j = 0; j < 8; j++) w[j] = *(const double2_t*)(p + j * 128 + (lane & 15) * 2);
    }
    long long c1 = clock64();
    if (MODE == 2 || MODE == 4) {
#pragma unroll
      for (int j = 0; j < 8; j++) acc += w[j].x + w[j].y;
    } else {
      if (MODE != 3 || lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; j++) acc += v[j];
      }
    }
    long long c2 = clock64();
    t_issue += c1 - c0;
    t_total += c2 - c0;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t_issue; cyc[1] = t_total; }
}
template <int MODE> void run(const char* name, int ninstr, int blocks, int threads, const double* in, double* out, long long* cyc) {
  int iters = 200;
  k<MODE><<<blocks, threads>>>(in, out, cyc, iters); hipDeviceSynchronize();
  k<MODE><<<blocks, threads>>>(in, out, cyc, iters); hipDeviceSynchronize();
  long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-44s blocks %4d x %3d thr: issue %.1f cyc/instr, issue+wait %.1f cyc/instr\n", name, blocks, threads, (double)h[0] / iters / ninstr, (double)h[1] / iters / ninstr);
}
int main() {
  double *in, *out; long long* cyc;
  size_t n = (size_t)1024 * 3 * (1 << 16) + (1 << 22);
  hipMalloc(&in, n * 8); hipMemset(in, 0, n * 8); hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&cyc, 16);
  for (int cfg = 0; cfg < 3; cfg++) {
    int blocks = cfg == 0 ? 1 : 256, threads = cfg == 2 ? 192 : 64;
    run<0>("16 x dwordx2, 64 distinct lanes", 16, blocks, threads, in, out, cyc);
    run<1>("16 x dwordx2, replicated x4", 16, blocks, threads, in, out, cyc);
    run<2>("8 x dwordx4, distinct", 8, blocks, threads, in, out, cyc);
    run<4>("8 x dwordx4, replicated x4", 8, blocks, threads, in, out, cyc);
    run<3>("16 x dwordx2, 16 active lanes", 16, blocks, threads, in, out, cyc);
  }
  return 0;
}
