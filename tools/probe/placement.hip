// Where does the dispatcher put workgroups and their waves?  (measurement helper)
// Each wave records HW_REG_HW_ID and HW_REG_XCC_ID plus start/end s_memrealtime (100 MHz, chip-global).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#include <algorithm>
__global__ void k_place(unsigned* out, int spin) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  unsigned long long t0, t1;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  lds[threadIdx.x] = a;
  __syncthreads();
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) {
    const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    out[w * 4 + 0] = hw;
    out[w * 4 + 1] = xcc;
    out[w * 4 + 2] = (unsigned)t0;
    out[w * 4 + 3] = (unsigned)(t1 - t0) + (lds[3] == 12345.f);
  }
}
static void run(int grid, int block, int lds, int spin) {
  const int nw = grid * block / 64;
  unsigned* d;
  hipMalloc(&d, nw * 16);
  hipFuncSetAttribute((const void*)k_place, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  k_place<<<grid, block, lds>>>(d, spin);
  hipDeviceSynchronize();
  k_place<<<grid, block, lds>>>(d, spin);
  hipDeviceSynchronize();
  std::vector<unsigned> h(nw * 4);
  hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost);
  printf("== grid %d block %d lds %d spin %d\n", grid, block, lds, spin);
  unsigned tmin = ~0u;
  for (int w = 0; w < nw; ++w) tmin = std::min(tmin, h[w * 4 + 2]);
  const int wpb = block / 64;
  // print first 24 blocks in detail
  for (int b = 0; b < std::min(grid, 40); ++b) {
    printf("blk %4d:", b);
    for (int w = 0; w < wpb; ++w) {
      const unsigned hw = h[(b * wpb + w) * 4], xcc = h[(b * wpb + w) * 4 + 1] & 0xf;
      printf(" [x%u se%u sh%u cu%2u simd%u slot%u t%u]", xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3, hw & 15, h[(b * wpb + w) * 4 + 2] - tmin);
    }
    printf("\n");
  }
  // census: blocks per CU, and for each CU the list of block ids
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < grid; ++b) {
    const unsigned hw = h[(b * wpb) * 4], xcc = h[(b * wpb) * 4 + 1] & 0xf;
    cu[(xcc << 16) | (hw & 0xff00)].push_back(b);
  }
  std::map<int, int> hist;
  for (auto& kv : cu) hist[(int)kv.second.size()]++;
  printf("CUs used %zu; blocks-per-CU histogram:", cu.size());
  for (auto& kv : hist) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
  int shown = 0;
  for (auto& kv : cu) {
    if (shown++ >= 6) break;
    printf("  cu %06x:", kv.first);
    for (int b : kv.second) printf(" %d", b);
    printf("\n");
  }
  // do waves w and w+4 of a block share a SIMD?  are a block's first 4 waves on 4 distinct SIMDs?
  int same = 0, distinct = 0;
  std::map<int, int> start_hist;
  for (int b = 0; b < grid; ++b) {
    unsigned s[16];
    for (int w = 0; w < wpb; ++w) s[w] = (h[(b * wpb + w) * 4] >> 4) & 3;
    if (wpb >= 4 && ((1u << s[0]) | (1u << s[1]) | (1u << s[2]) | (1u << s[3])) == 15) ++distinct;
    if (wpb >= 8 && s[0] == s[4] && s[1] == s[5] && s[2] == s[6] && s[3] == s[7]) ++same;
    start_hist[s[0] * 1000 + s[1] * 100 + s[2] * 10 + s[3]]++;
  }
  printf("blocks whose waves 0-3 sit on 4 distinct SIMDs: %d/%d; waves w,w+4 on the same SIMD: %d\n", distinct, grid, same);
  printf("SIMD sequence of waves 0..3 histogram:");
  for (auto& kv : start_hist) printf(" %04d:%d", kv.first, kv.second);
  printf("\n");
  hipFree(d);
}
int main() {
  run(1024, 256, 39 * 1024, 20000);
  run(512, 512, 65 * 1024, 20000);
  run(1024, 256, 52 * 1024, 20000);
  return 0;
}
