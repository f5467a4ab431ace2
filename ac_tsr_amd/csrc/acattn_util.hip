// Small streaming helpers of the training step.
//
// acattn_sum_rows: out[bt, c] = sum_r x[bt, r, c].  The step needs ~60 such reductions (bias gradients, split-K
// partial slabs, per-(b,head) parameter partials, per-head gate gradients); torch's generic reduce kernel takes
// ~10 us for each of them regardless of size, which made them the largest torch item of the step.
#include "acattn_common.h"

namespace {

// Block = CT column threads (4 floats each) x RL row lanes; a block reduces `rows_per_chunk` rows of a
// 4*CT-column strip, folds its row lanes through LDS and stores (one chunk) or atomically adds (several).
template <bool ATOMIC>
__global__ void __launch_bounds__(256) sum_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C,
                                                       int rows_per_chunk, int CT) {
  __shared__ f4 red[256];
  const int RL = 256 / CT;
  const int ct = threadIdx.x % CT, rl = threadIdx.x / CT;
  const int c0 = (blockIdx.x * CT + ct) * 4;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  const bool vec = (C & 3) == 0;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c0 < C) {
    const float* base = x + (size_t)blockIdx.z * R * C + c0;
    if (vec) {
      int r = r0 + rl;
      for (; r + 3 * RL < r1; r += 4 * RL) {  // 4 independent loads in flight
        const f4 a = *(const f4*)(base + (size_t)r * C), b = *(const f4*)(base + (size_t)(r + RL) * C);
        const f4 c = *(const f4*)(base + (size_t)(r + 2 * RL) * C), d = *(const f4*)(base + (size_t)(r + 3 * RL) * C);
        acc += (a + b) + (c + d);
      }
      for (; r < r1; r += RL) acc += *(const f4*)(base + (size_t)r * C);
    } else {
      for (int r = r0 + rl; r < r1; r += RL)
        for (int e = 0; e < 4 && c0 + e < C; ++e) acc[e] += base[(size_t)r * C + e];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rl == 0 && c0 < C) {
    for (int k = 1; k < RL; ++k) acc += red[k * CT + ct];
    float* o = out + (size_t)blockIdx.z * C + c0;
    for (int e = 0; e < 4 && c0 + e < C; ++e) {
      if (ATOMIC)
        atomicAdd(o + e, acc[e]);
      else
        o[e] = acc[e];
    }
  }
}

}  // namespace

int acattn_launch_sum_rows(const float* x, float* out, int batch, int R, int C, hipStream_t stream) {
  const int c4 = (C + 3) / 4;
  int CT = 256;
  while (CT > 1 && CT / 2 >= c4) CT /= 2;  // smallest power of two >= c4, capped at 256
  const int col_groups = (c4 + CT - 1) / CT;
  // one chunk: every output element is written by exactly one workgroup (no atomics, no zero-fill).  Callers that
  // need more parallelism over a long R split it themselves into [batch * s, R / s, C] and reduce twice (ops.sum_rows)
  const dim3 grid(col_groups, 1, batch);
  hipLaunchKernelGGL(sum_rows_kernel<false>, grid, dim3(256), 0, stream, x, out, R, C, R, CT);
  return (int)hipGetLastError();
}
