// LDS throughput per CU for the operations of the attention backward's key side: ds_add_f32 (no return) against
// ds_write_b32 / ds_read_b32, 8 waves per CU (2 workgroups of 256), conflict-free addresses.  (measurement helper)
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  __shared__ float lds[64 * 36 * 3 + 256];
  for (int i = threadIdx.x; i < 64 * 36 * 3; i += 256) lds[i] = 0.f;
  __syncthreads();
  const int lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  float* p = lds + (4 * g) * 36 + c;  // the accumulator pattern of key_side: rows 4g + r, stride 36, columns c
  float v = threadIdx.x * 1e-3f, acc = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (MODE == 0) atomicAdd(p + (j & 3) * 36 + 16 * (j >> 2), v);
      else if (MODE == 1) p[(j & 3) * 36 + 16 * (j >> 2)] = v;
      else acc += p[(j & 3) * 36 + 16 * (j >> 2)];
    }
    if (MODE == 1) asm volatile("" ::: "memory");
    if (MODE == 2) asm volatile("" :: "v"(acc) : "memory");
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x] + acc;
}
template <int MODE>
float run(float* out, int iters, int wgs) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<wgs, 256>>>(out, iters); hipDeviceSynchronize();
  hipEventRecord(e0); k<MODE><<<wgs, 256>>>(out, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1000;
}
int main() {
  float* out; hipMalloc(&out, 2048 * 256 * 4);
  const int it = 2000;  // 16000 LDS instructions per wave
  for (int wgs : {256, 512}) {
    const double n = 16000.0 * (wgs / 256) * 4;  // LDS instructions per CU
    const float a = run<0>(out, it, wgs), w = run<1>(out, it, wgs), r = run<2>(out, it, wgs);
    printf("%d workgroups/CU: ds_add_f32 %.0f us (%.1f cycles/instr/CU at 2.4 GHz) | ds_write_b32 %.0f us (%.1f) | ds_read_b32 %.0f us (%.1f)\n",
           wgs / 256, a, a * 2400 / n, w, w * 2400 / n, r, r * 2400 / n);
  }
  return 0;
}
