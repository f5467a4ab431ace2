// Do fp32 MFMA and VALU streams of DIFFERENT waves on one SIMD overlap?  (measurement helper)
// 2 or 4 blocks per CU (LDS-forced); role by blockIdx/256 (blocks b, b+256, .. share a CU): MFMA-only or VALU-only.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(float* out, int iters, int role_mask) {
  extern __shared__ float lds[];
  const int slot = blockIdx.x / 256;
  const bool mfma = (role_mask >> slot) & 1;
  float a[8]; f4 m[2] = {{1,2,3,4},{4,3,2,1}};
  for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  const float b = 1.0001f, c = 0.001f;
  if (mfma) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[j & 1]) : "v"(b), "v"(c));
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  s += m[0][0] + m[1][1];
  lds[threadIdx.x] = s;
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
static float run(int slots, int role_mask, int iters, float* out) {
  const int lds = (156 * 1024 / slots) & ~255;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<<<256 * slots, 256, lds>>>(out, iters, role_mask); hipDeviceSynchronize();
  hipEventRecord(e0); k<<<256 * slots, 256, lds>>>(out, iters, role_mask); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000;
}
int main() {
  float* out; hipMalloc(&out, 1024 * 256 * 4);
  const int it = 4000;  // MFMA wave: 32000 MFMA (x13.6ns = 435us);  VALU wave: 128000 fma
  printf("2 slots: MFMA+MFMA %.0f us | VALU+VALU %.0f us | MFMA+VALU %.0f us\n", run(2, 3, it, out), run(2, 0, it, out), run(2, 1, it, out));
  printf("4 slots: 4xMFMA %.0f | 4xVALU %.0f | 2+2 %.0f | 1 MFMA + 3 VALU %.0f | 3 MFMA + 1 VALU %.0f\n", run(4, 15, it, out), run(4, 0, it, out), run(4, 5, it, out), run(4, 1, it, out), run(4, 7, it, out));
  printf("1 slot: MFMA %.0f | VALU %.0f\n", run(1, 1, it, out), run(1, 0, it, out));
  return 0;
}
