// Does private (scratch) memory stay private at full occupancy?  Every lane fills a dynamically indexed local array
// (forced into scratch), idles, and checks it.  tools/probe (measurement helper, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
template <int WAVES>
__global__ void __launch_bounds__(64, WAVES) k_scratch(unsigned* bad, int n, int spin) {
  volatile unsigned a[24];
  const unsigned id = blockIdx.x * 64u + threadIdx.x;
  for (int i = 0; i < 24; ++i) a[(i * 7 + n) % 24] = id * 31u + i;
  float x = (float)id;
  for (int i = 0; i < spin; ++i) x = fmaf(x, 1.000001f, 0.5f);
  unsigned errs = 0;
  for (int i = 0; i < 24; ++i) errs += a[(i * 7 + n) % 24] != id * 31u + i;
  if (errs || x == 12345.f) atomicAdd(bad, errs ? 1u : 0u);
}
int main() {
  unsigned* d; CK(hipMalloc(&d, 4));
  for (int grid : {256, 1024, 2048, 4096, 8192, 16384}) {
    for (int w : {1, 2, 4}) {
      CK(hipMemset(d, 0, 4));
      if (w == 1) hipLaunchKernelGGL(k_scratch<1>, dim3(grid), dim3(64), 0, 0, d, 3, 2000);
      if (w == 2) hipLaunchKernelGGL(k_scratch<2>, dim3(grid), dim3(64), 0, 0, d, 3, 2000);
      if (w == 4) hipLaunchKernelGGL(k_scratch<4>, dim3(grid), dim3(64), 0, 0, d, 3, 2000);
      CK(hipDeviceSynchronize());
      unsigned h; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
      printf("grid %6d x 64, launch_bounds waves %d: lanes with a corrupted private array: %u\n", grid, w, h);
    }
  }
  return 0;
}
