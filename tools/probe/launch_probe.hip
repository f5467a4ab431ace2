// Launch + dispatch cost of an (almost) empty kernel against the grid shape: what a one-generation launch of N waves
// pays before any wave does useful work.  tools/probe (measurement helper, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); return 1; } } while (0)
template <int VG>
__global__ void __launch_bounds__(1024) k_empty(float* out, int n) {
  if (n == -12345) out[threadIdx.x] = 1.f;
}
// a kernel that holds many registers (launch_bounds 64 threads, 4 waves/SIMD -> 128 VGPRs) and does ~K dependent FMAs
template <int K>
__global__ void __launch_bounds__(64, 4) k_work(float* out, int n, float a) {
  float x[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) x[i] = a + i;
  for (int it = 0; it < K; ++it) {
#pragma unroll
    for (int i = 0; i < 32; ++i) x[i] = fmaf(x[i], a, x[(i + 1) & 31]);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) s += x[i];
  if (n == -12345) out[threadIdx.x] = s;
}
int main() {
  float* d; CK(hipMalloc(&d, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](auto launch, const char* name) {
    std::vector<float> us;
    for (int r = -1; r < 8; ++r) {
      hipEventRecord(e0, st);
      for (int i = 0; i < 100; ++i) launch();
      hipEventRecord(e1, st); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r >= 0) us.push_back(ms * 10.f);
    }
    std::sort(us.begin(), us.end());
    printf("%-40s median %.2f us  min %.2f\n", name, us[us.size() / 2], us[0]);
  };
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(1), dim3(64), 0, st, d, 0); }, "empty 1 x 64");
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(256), dim3(64), 0, st, d, 0); }, "empty 256 x 64");
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(1024), dim3(64), 0, st, d, 0); }, "empty 1024 x 64");
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(4096), dim3(64), 0, st, d, 0); }, "empty 4096 x 64");
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(2048), dim3(128), 0, st, d, 0); }, "empty 2048 x 128");
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(1024), dim3(256), 0, st, d, 0); }, "empty 1024 x 256");
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(512), dim3(512), 0, st, d, 0); }, "empty 512 x 512");
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(256), dim3(1024), 0, st, d, 0); }, "empty 256 x 1024");
  time([&] { hipLaunchKernelGGL(k_empty<0>, dim3(8192), dim3(64), 0, st, d, 0); }, "empty 8192 x 64");
  time([&] { hipLaunchKernelGGL(k_work<8>, dim3(4096), dim3(64), 0, st, d, 0, 1.0001f); }, "work8 (256 fma) 4096 x 64");
  time([&] { hipLaunchKernelGGL(k_work<32>, dim3(4096), dim3(64), 0, st, d, 0, 1.0001f); }, "work32 (1024 fma) 4096 x 64");
  time([&] { hipLaunchKernelGGL(k_work<64>, dim3(4096), dim3(64), 0, st, d, 0, 1.0001f); }, "work64 (2048 fma) 4096 x 64");
  time([&] { hipLaunchKernelGGL(k_work<64>, dim3(2048), dim3(64), 0, st, d, 0, 1.0001f); }, "work64 (2048 fma) 2048 x 64");
  time([&] { hipLaunchKernelGGL(k_work<64>, dim3(1024), dim3(64), 0, st, d, 0, 1.0001f); }, "work64 (2048 fma) 1024 x 64");
  time([&] { hipLaunchKernelGGL(k_work<64>, dim3(8192), dim3(64), 0, st, d, 0, 1.0001f); }, "work64 (2048 fma) 8192 x 64");
  return 0;
}
