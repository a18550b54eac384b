// semantics of the DPP controls used by the 16-lanes-per-trajectory backward kernel (backward_hex.hpp)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
__global__ void k(int* out) {
  const int l = threadIdx.x;
  out[0 * 64 + l] = dpp<0x124>(l);   // row_ror:4
  out[1 * 64 + l] = dpp<0x128>(l);   // row_ror:8
  out[2 * 64 + l] = dpp<0x12C>(l);   // row_ror:12
  out[3 * 64 + l] = dpp<0x150>(l);   // row_newbcast:0
  out[4 * 64 + l] = dpp<0x154>(l);   // row_newbcast:4
  out[5 * 64 + l] = dpp<0x15D>(l);   // row_newbcast:13
  out[6 * 64 + l] = dpp<0xB1>(l);    // quad_perm [1,0,3,2]
  out[7 * 64 + l] = dpp<0x4E>(l);    // quad_perm [2,3,0,1]
}
int main() {
  int* d; hipMalloc(&d, 8 * 64 * 4);
  k<<<1, 64>>>(d);
  int h[8 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* nm[] = {"row_ror:4", "row_ror:8", "row_ror:12", "row_newbcast:0", "row_newbcast:4", "row_newbcast:13", "quad_perm[1,0,3,2]", "quad_perm[2,3,0,1]"};
  for (int i = 0; i < 8; i++) { printf("%-20s", nm[i]); for (int l = 16; l < 32; l++) printf(" %2d", h[i * 64 + l]); printf("\n"); }
  return 0;
}
