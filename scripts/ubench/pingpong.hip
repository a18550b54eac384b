// What does a hand-over between two wavefronts of a workgroup (different SIMDs) cost through LDS?  Wavefront 0 writes a payload and a
// sequence number, wavefront 1 spins on it, reads the payload, writes a reply and its own sequence number, wavefront 0 spins on that:
// s_memtime cycles per ROUND TRIP (two hand-overs), with and without 2 other wavefronts busy on the remaining SIMDs.
// hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip && ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k(long long* out, double* sink, int rounds, int busy) {
  __shared__ double req[5 * 16], rep[3 * 16];
  __shared__ int req_seq, rep_seq;
  if (threadIdx.x == 0) { req_seq = 0; rep_seq = 0; }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l = lane >> 2;
  double acc = lane;
  if (wave == 0) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int n = 1; n <= rounds; n++) {
      if ((lane & 3) == 0)
        for (int e = 0; e < 5; e++) req[e * 16 + l] = acc + e;
      __hip_atomic_store(&req_seq, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      while (__hip_atomic_load(&rep_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != n) {}
      acc += rep[l] + rep[16 + l] + rep[32 + l];
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[0] = (t1 - t0) / rounds;
  } else if (wave == 1) {
    for (int n = 1; n <= rounds; n++) {
      while (__hip_atomic_load(&req_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != n) {}
      double s = 0;
      for (int e = 0; e < 5; e++) s += req[e * 16 + l];
      if ((lane & 3) == 0)
        for (int e = 0; e < 3; e++) rep[e * 16 + l] = s * (e + 1);
      __hip_atomic_store(&rep_seq, n, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  } else if (busy) {  // the other two SIMDs: FMA chains with some LDS traffic, until wavefront 0 is through
    volatile int* done = &rep_seq;
    while (*done < rounds) {
      for (int i = 0; i < 64; i++) acc = __builtin_fma(acc, 1.0000001, 0.5);
      req[80 - 1 - (lane & 7)] = acc;  // (unused tail)
    }
  }
  sink[threadIdx.x] = acc;
}
int main() {
  long long* out; double* sink;
  (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 256 * 8);
  for (int busy = 0; busy < 2; busy++) {
    for (int rep = 0; rep < 2; rep++) { k<<<1, 256>>>(out, sink, 20000, busy); (void)hipDeviceSynchronize(); }
    long long h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("LDS ping-pong between two wavefronts, other SIMDs %s: %lld cycles per round trip (request 5 x 16 doubles, reply 3 x 16)\n", busy ? "busy" : "idle", h);
  }
  return 0;
}
