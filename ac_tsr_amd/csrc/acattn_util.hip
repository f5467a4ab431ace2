// Small streaming helpers of the training step.
//
// acattn_sum_rows: out[bt, c] = sum_r x[bt, r, c].  The step needs ~60 such reductions (bias gradients, split-K
// partial slabs, per-(b,head) parameter partials, per-head gate gradients); torch's generic reduce kernel takes
// ~10 us for each of them regardless of size, which made them the largest torch item of the step.
#include <algorithm>

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

// ---- mask penalty: || 1 - M ||_2 over the whole attack-mask tensor (acsasrec.py:131-137, acbert4rec.py:229-232) ----
// torch spells it rsub -> norm (two 10 MB round trips) and, backward, four elementwise kernels; here the forward is
// one read of M (+ a one-workgroup fold that also takes the square root) and the backward one read + one write.
constexpr int kPenaltyGrid = 1024;

__global__ void __launch_bounds__(256) penalty_partial_kernel(const float* __restrict__ m, const int64_t n,
                                                              float* __restrict__ part) {
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f4 d = 1.0f - *(const f4*)(m + 4 * i);
    acc += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail of a length that is not a multiple of 4
    const float d = 1.0f - m[(n4 << 2) + threadIdx.x];
    acc += d * d;
  }
  __shared__ float red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ void __launch_bounds__(256) penalty_finish_kernel(const float* __restrict__ part, const int n_part,
                                                             float* __restrict__ norm) {
  __shared__ float red[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n_part; i += 256) acc += part[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) norm[0] = sqrtf(red[0]);
}

// d_m = d_norm * (m - 1) / norm     (d ||1 - m|| / d m = -(1 - m) / ||1 - m||; 0 when the norm is 0, like torch)
__global__ void __launch_bounds__(256) penalty_bwd_kernel(const float* __restrict__ m, const float* __restrict__ norm,
                                                          const float* __restrict__ d_norm, const int64_t n,
                                                          float* __restrict__ d_m) {
  const float nv = norm[0];
  const float k = nv > 0.f ? d_norm[0] / nv : 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    *(f4*)(d_m + 4 * i) = (*(const f4*)(m + 4 * i) - 1.0f) * k;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t j = (n4 << 2) + threadIdx.x;
    d_m[j] = (m[j] - 1.0f) * k;
  }
}

}  // namespace

int64_t acattn_penalty_ws_floats() { return kPenaltyGrid; }

int acattn_launch_penalty_fwd(const float* m, int64_t n, float* ws, float* norm, hipStream_t stream) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, kPenaltyGrid));
  hipLaunchKernelGGL(penalty_partial_kernel, dim3(grid), dim3(256), 0, stream, m, n, ws);
  hipLaunchKernelGGL(penalty_finish_kernel, dim3(1), dim3(256), 0, stream, ws, grid, norm);
  return (int)hipGetLastError();
}

int acattn_launch_penalty_bwd(const float* m, const float* norm, const float* d_norm, int64_t n, float* d_m,
                              hipStream_t stream) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, 2048));
  hipLaunchKernelGGL(penalty_bwd_kernel, dim3(grid), dim3(256), 0, stream, m, norm, d_norm, n, d_m);
  return (int)hipGetLastError();
}

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
