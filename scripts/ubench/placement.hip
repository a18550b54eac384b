// Where do the wavefronts of co-resident workgroups land?  (HW_ID / LDS_ALLOC / XCC_ID per wavefront)
// hipcc --offload-arch=gfx950 -O3 -o placement placement.hip && ./placement [lds_kb] [blocks]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg((id) | ((off) << 6) | (((size)-1) << 11))
extern __shared__ double dyn[];
__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  const int wave = threadIdx.x >> 6;
  const unsigned hw = GETREG(4, 0, 32), lds = GETREG(6, 0, 32), xcc = GETREG(20, 0, 32);
  long long t0 = __builtin_amdgcn_s_memtime();
  dyn[threadIdx.x] = threadIdx.x;
  while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(10);  // keep the block resident so that others pile up next to it
  if ((threadIdx.x & 63) == 0) {
    unsigned* o = out + (blockIdx.x * 4 + wave) * 4;
    o[0] = hw; o[1] = lds; o[2] = xcc; o[3] = (unsigned)dyn[threadIdx.x];
  }
}
int main(int argc, char** argv) {
  const int lds_kb = argc > 1 ? atoi(argv[1]) : 72, blocks = argc > 2 ? atoi(argv[2]) : 512;
  unsigned* out; hipMalloc(&out, blocks * 64);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
  probe<<<blocks, 256, lds_kb * 1024>>>(out, 2000000);
  hipError_t e = hipDeviceSynchronize(); fprintf(stderr, "%s\n", hipGetErrorString(e));
  std::vector<unsigned> h(blocks * 16); hipMemcpy(h.data(), out, blocks * 64, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> per_cu;
  int distinct = 0;
  for (int b = 0; b < blocks; b++) {
    unsigned simds = 0;
    for (int w = 0; w < 4; w++) simds |= 1u << ((h[(b * 4 + w) * 4] >> 4) & 3);
    distinct += (simds == 0xF);
    const unsigned hw = h[b * 16], xcc = h[b * 16 + 2];
    per_cu[((xcc & 0xF) << 16) | (hw & 0xFF00)].push_back(b);  // cu_id[11:8], sh[12], se[15:13]
  }
  printf("blocks %d, lds %d KB: blocks whose 4 wavefronts sit on 4 distinct SIMDs: %d; distinct (xcc,se,sh,cu): %zu\n", blocks, lds_kb, distinct, per_cu.size());
  int shown = 0;
  for (auto& kv : per_cu) {
    if (shown++ >= 6) break;
    printf("cu key %06x:", kv.first);
    for (int b : kv.second) {
      printf("  [blk %d lds %08x simd/wave", b, h[b * 16 + 1]);
      for (int w = 0; w < 4; w++) printf(" %u/%u", (h[(b * 4 + w) * 4] >> 4) & 3, h[(b * 4 + w) * 4] & 15);
      printf("]");
    }
    printf("\n");
  }
  return 0;
}
