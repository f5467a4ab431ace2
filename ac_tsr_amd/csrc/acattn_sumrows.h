// acattn_sum_rows building blocks shared by acattn_util.hip (the reductions' own launches) and acattn_linear.hip (the
// deferred reductions of a backward walk, acattn_linear_wgrad_reduce_many).
#pragma once
#include "acattn_common.h"

namespace {

// Block = CT column threads (4 floats each) x RL row lanes; a block reduces `rows_per_chunk` rows of a
// 4*CT-column strip, folds its row lanes through LDS and stores (one chunk) or atomically adds (several).
template <bool ATOMIC>
__device__ __forceinline__ void sum_rows_block(const float* __restrict__ x, float* __restrict__ out, int R, int C,
                                               int rows_per_chunk, int CT, int col_group, int chunk, int bt) {
  __shared__ f4 red[256];
  const int RL = 256 / CT;
  const int ct = threadIdx.x % CT, rl = threadIdx.x / CT;
  const int c0 = (col_group * CT + ct) * 4;
  const int r0 = chunk * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  const bool vec = (C & 3) == 0;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  if (c0 < C) {
    const float* base = x + (size_t)bt * R * C + c0;
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
    float* o = out + (size_t)bt * C + c0;
    for (int e = 0; e < 4 && c0 + e < C; ++e) {
      if (ATOMIC)
        atomicAdd(o + e, acc[e]);
      else
        o[e] = acc[e];
    }
  }
}

struct SumRowsJob {
  const float* x;
  float* out;
  int R, C, CT, col_groups, n_wg;
};

// column threads per workgroup of one reduction (see sum_rows_block)
inline int pick_ct(int batch, int R, int C) {
  const int c4 = (C + 3) / 4;
  int CT = 256;
  while (CT > 1 && CT / 2 >= c4) CT /= 2;  // smallest power of two >= c4, capped at 256
  // few, narrow outputs over many rows (1,024 x 132 calibrator partials: one workgroup, 20 us): give a workgroup
  // fewer columns and more row lanes until about 32 workgroups share the reduction
  while (CT > 1 && (int64_t)batch * ((c4 + CT - 1) / CT) < 32 && R >= 4 * (256 / CT)) CT /= 2;
  return CT;
}
inline SumRowsJob make_job(const float* x, float* out, int batch, int R, int C) {
  SumRowsJob j;
  j.x = x, j.out = out, j.R = R, j.C = C;
  j.CT = pick_ct(batch, R, C);
  j.col_groups = ((C + 3) / 4 + j.CT - 1) / j.CT;
  j.n_wg = j.col_groups * batch;
  return j;
}

}  // namespace
