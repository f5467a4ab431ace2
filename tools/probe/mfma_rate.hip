// fp32 MFMA issue rate on gfx950: CH independent accumulator chains per wave, W waves per SIMD, all 256 CUs.
//   hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/probe/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CH>
__global__ void __launch_bounds__(64) rate_kernel(float* out, int iters, float a0) {
  f4 acc[CH];
  for (int k = 0; k < CH; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + threadIdx.x, b = a0 * 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int k = 0; k < CH; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < CH; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == 12345.678f) out[0] = s;
}
template <int CH>
void run(float* d, int waves_per_simd) {
  const int iters = 4000 / CH, blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<CH>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<CH>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n_per_simd = (double)iters * 16 * CH * waves_per_simd;
  printf("chains %d waves/SIMD %d: %8.1f us, %6.2f ns per MFMA per SIMD, %6.1f TFLOP/s\n", CH, waves_per_simd, ms * 1e3,
         ms * 1e6 / n_per_simd, n_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
  float* d;
  hipMalloc(&d, 1024);
  for (int w : {1, 2, 4}) {
    run<1>(d, w);
    run<2>(d, w);
    run<4>(d, w);
  }
  return 0;
}
