// Issue-rate microbenchmark (measurement helper, not product): cycles per wave-instruction on one SIMD with
// 1, 2, 4 or 8 resident waves per SIMD, for the instruction kinds the forward kernel is made of.
// Each wave runs ITER x 8 independent instructions of one kind and stamps s_memtime around the loop.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void __launch_bounds__(256) k_rate(float* out, unsigned long long* cyc, int iters, float seedv) {
  float a[8];
  f2 p[8];
  f4 m[4];
  unsigned u[8];
  extern __shared__ float lds[];  // sized by the host so that exactly `wps` blocks fit one CU
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * seedv;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = seedv * (threadIdx.x + i + 1);
    p[i] = f2{a[i], a[i] * 0.5f};
    u[i] = (unsigned)(threadIdx.x * 977 + i * 131 + 7);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) m[i] = f4{a[i], a[i + 1], a[i + 2], a[i + 3]};
  const float b = seedv * 0.999f, c = seedv * 0.001f;
  const f2 b2 = f2{b, b}, c2 = f2{c, c};
  const unsigned lofs = (threadIdx.x & 63) * 16;
  const unsigned long long msk = 0x5555aaaa5555aaaaull ^ (unsigned long long)iters;
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if constexpr (OP == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
      REP8(X)
#undef X
    } else if constexpr (OP == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      REP8(X)
#undef X
    } else if constexpr (OP == 3) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      REP8(X)
#undef X
    } else if constexpr (OP == 4) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
      REP8(X)
#undef X
    } else if constexpr (OP == 5) {
#define X(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
      REP8(X)
#undef X
    } else if constexpr (OP == 6) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(msk));
      REP8(X)
#undef X
    } else if constexpr (OP == 7) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 4) & 7]));
      X(0) X(1) X(2) X(3) X(0) X(1) X(2) X(3)
#undef X
    } else if constexpr (OP == 8) {  // independent fp32 MFMA 16x16x4, 4 accumulators
#define X(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[i & 3]) : "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if constexpr (OP == 9) {  // dependent chain on one accumulator
#define X(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[0]) : "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if constexpr (OP == 10) {  // two chains
#define X(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[i & 1]) : "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if constexpr (OP == 11) {
#define X(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
      REP8(X)
#undef X
    } else if constexpr (OP == 12) {
#define X(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
      REP8(X)
#undef X
    } else if constexpr (OP == 13) {
#define X(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
      REP8(X)
#undef X
    } else if constexpr (OP == 14) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
      REP8(X)
#undef X
    } else if constexpr (OP == 15) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP8(X)
#undef X
    } else if constexpr (OP == 16) {  // ds_read_b128, waited every 8
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*1024" : "=v"(m[i & 3]) : "v"(lofs));
      REP8(X)
#undef X
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (OP == 17) {  // mixed: 1 MFMA + 6 VALU (fma) per group
      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(m[0]) : "v"(b), "v"(c));
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      X(0) X(1) X(2) X(3) X(4) X(5) X(6)
#undef X
    } else if constexpr (OP == 18) {
#define X(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
      REP8(X)
#undef X
    } else if constexpr (OP == 19) {
#define X(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
      REP8(X)
#undef X
    } else if constexpr (OP == 20) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
      REP8(X)
#undef X
    } else if constexpr (OP == 21) {
#define X(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      REP8(X)
#undef X
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i] + p[i][0] + p[i][1] + (float)u[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += m[i][0] + m[i][1] + m[i][2] + m[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, unsigned long long* cyc, int iters) {
  printf("%-28s", name);
  for (int wps : {1, 2, 4, 8}) {  // blocks per CU = waves per SIMD (256-thread blocks: one wave per SIMD each)
    const int grid = 256 * wps;
    const size_t lds = ((156 * 1024 / wps) & ~255);  // forces an even spread: wps blocks per CU, no more
    hipFuncSetAttribute((const void*)k_rate<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_rate<OP><<<grid, 256, lds>>>(out, cyc, iters, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_rate<OP><<<grid, 256, lds>>>(out, cyc, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    // cycles of SIMD time per wave-instruction = wave elapsed / (iters*8) / wps  (wps waves share the SIMD)
    printf("  w%d: %5.2f cyc/inst/SIMD, wave %5.1f (%4.0f us, %.2f ns/inst/SIMD)", wps, med / (iters * 8.0) / wps, med / (iters * 8.0), ms * 1000,
           ms * 1e6 / (iters * 8.0) / wps);
  }
  printf("\n");
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMalloc(&cyc, 256 * 8 * 4 * 8);
  const int it = 4000;
  run<0>("v_fma_f32", out, cyc, it);
  run<1>("v_pk_fma_f32", out, cyc, it);
  run<14>("v_pk_mul_f32", out, cyc, it);
  run<2>("v_exp_f32", out, cyc, it);
  run<3>("v_rcp_f32", out, cyc, it);
  run<12>("v_log_f32", out, cyc, it);
  run<13>("v_sin_f32", out, cyc, it);
  run<21>("v_sqrt_f32", out, cyc, it);
  run<4>("v_mul_lo_u32", out, cyc, it);
  run<11>("v_mad_u32_u24", out, cyc, it);
  run<5>("v_xor_b32", out, cyc, it);
  run<18>("v_lshl_add_u32", out, cyc, it);
  run<6>("v_cndmask_b32", out, cyc, it);
  run<20>("v_cmp_lt_f32", out, cyc, it);
  run<19>("v_cvt_f32_u32", out, cyc, it);
  run<15>("v_max3_f32", out, cyc, it);
  run<7>("v_permlane32_swap", out, cyc, it);
  run<8>("mfma16x16x4 f32 indep x4", out, cyc, it);
  run<10>("mfma16x16x4 f32 2 chains", out, cyc, it);
  run<9>("mfma16x16x4 f32 1 chain", out, cyc, it);
  run<17>("1 mfma + 7 fma", out, cyc, it);
  run<16>("ds_read_b128 (x8, wait)", out, cyc, it);
  return 0;
}
