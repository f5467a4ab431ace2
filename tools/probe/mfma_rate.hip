// fp32 MFMA issue rate on gfx950: CH independent accumulator chains per wave, W waves per SIMD, all 256 CUs;
// operands either two constants per lane (RANDOM = 0) or 2 x 16 registers of random data per lane (RANDOM = 1: the
// power-relevant case).   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_rate.hip -o mfma_rate 2>/dev/null
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int CH, bool RANDOM>
__global__ void __launch_bounds__(64) rate_kernel(float* out, const float* src, int iters, float a0) {
  f4 acc[CH];
  for (int k = 0; k < CH; ++k) acc[k] = f4{0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
  for (int u = 0; u < 16; ++u) {
    a[u] = RANDOM ? src[(blockIdx.x & 63) * 2048 + u * 64 + threadIdx.x] : a0 + threadIdx.x;
    b[u] = RANDOM ? src[(blockIdx.x & 63) * 2048 + 1024 + u * 64 + threadIdx.x] : a0 * 0.5f;
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int k = 0; k < CH; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[k], 0, 0, 0);
  }
  float s = 0.f;
  for (int k = 0; k < CH; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == 12345.678f) out[0] = s;
}
template <int CH, bool RANDOM>
void run(float* d, const float* src, int waves_per_simd) {
  const int iters = 4000 / CH, blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((rate_kernel<CH, RANDOM>), dim3(blocks), dim3(64), 0, 0, d, src, iters, 1.0f);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL((rate_kernel<CH, RANDOM>), dim3(blocks), dim3(64), 0, 0, d, src, iters, 1.0f);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double n_per_simd = (double)iters * 16 * CH * waves_per_simd;
  printf("%s chains %d waves/SIMD %d: %8.1f us, %6.2f ns per MFMA per SIMD, %6.1f TFLOP/s\n", RANDOM ? "random  " : "constant", CH,
         waves_per_simd, ms * 1e3, ms * 1e6 / n_per_simd, n_per_simd * 1024 * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
  float *d, *src;
  (void)hipMalloc(&d, 1024);
  std::vector<float> h(64 * 2048);
  srand(1);
  for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  (void)hipMalloc(&src, h.size() * 4);
  (void)hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  for (int w : {1, 2}) {
    run<2, false>(d, src, w);
    run<2, true>(d, src, w);
    run<1, true>(d, src, w);
  }
  return 0;
}
