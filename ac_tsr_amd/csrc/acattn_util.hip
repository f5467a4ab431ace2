// Small streaming helpers of the training step.
//
// acattn_sum_rows: out[bt, c] = sum_r x[bt, r, c].  The step needs ~60 such reductions (bias gradients, split-K
// partial slabs, per-(b,head) parameter partials, per-head gate gradients); torch's generic reduce kernel takes
// ~10 us for each of them regardless of size, which made them the largest torch item of the step.
#include <algorithm>

#include <stdlib.h>

#include "acattn_common.h"
#include "acattn_sumrows.h"

namespace {

template <bool ATOMIC>
__global__ void __launch_bounds__(256) sum_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C,
                                                       int rows_per_chunk, int CT) {
  sum_rows_block<ATOMIC>(x, out, R, C, rows_per_chunk, CT, blockIdx.x, blockIdx.y, blockIdx.z);
}

// two independent reductions in one launch (the attention backward's parameter partials and its per-head gate gradient
// are produced together and summed together: one launch at the ~5 us floor instead of two)
__global__ void __launch_bounds__(256) sum_rows_pair_kernel(const SumRowsJob a, const SumRowsJob b) {
  const bool first = (int)blockIdx.x < a.n_wg;  // (uniform per workgroup)
  const SumRowsJob& j = first ? a : b;
  const int w = first ? blockIdx.x : blockIdx.x - a.n_wg;
  sum_rows_block<false>(j.x, j.out, j.R, j.C, j.R, j.CT, w % j.col_groups, 0, w / j.col_groups);
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

// The attacked loss assembled by ONE workgroup (acsasrec.py:129-137):
//     loss = -mean_b row_loss_b + weight * mean_l sqrt(sum part[l][:])
// out[0] = loss, out[1] = mean row loss, out[2 + l] = ||1 - M_l||.  `scale_buf` (the CE direction, [n_scale] floats) is
// multiplied by -1/B by the launch's other workgroups so that the backward of the cross-entropy term is one product with d loss.
__global__ void __launch_bounds__(256) attacked_loss_finish_kernel(const float* __restrict__ row_loss, const int B,
                                                                   const float* __restrict__ part, const int n_masks,
                                                                   const int n_part, const int part_stride,
                                                                   const float weight, float* __restrict__ out,
                                                                   float* __restrict__ scale_buf, const int n_scale) {
  const float k = -1.0f / (float)B;
  if (blockIdx.x > 0) {  // workgroups 1.. scale the direction, workgroup 0 assembles the loss
    for (int i = (blockIdx.x - 1) * 256 + threadIdx.x; i < n_scale / 4; i += (gridDim.x - 1) * 256)
      *(f4*)(scale_buf + 4 * i) = *(const f4*)(scale_buf + 4 * i) * k;
    if (blockIdx.x == 1 && threadIdx.x < (n_scale & 3)) scale_buf[(n_scale & ~3) + threadIdx.x] *= k;
    return;
  }
  __shared__ float red[256];
  auto block_sum = [&](float v) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  float acc = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) acc += row_loss[i];
  const float ce = block_sum(acc) / (float)B;
  float pen = 0.f;
  for (int l = 0; l < n_masks; ++l) {
    float a = 0.f;
    for (int i = threadIdx.x; i < n_part; i += 256) a += part[(size_t)l * part_stride + i];
    const float nv = sqrtf(block_sum(a));
    if (threadIdx.x == 0) out[2 + l] = nv;
    pen += nv;
  }
  if (threadIdx.x == 0) {
    out[0] = weight * (pen / (float)n_masks) - ce;
    out[1] = ce;
  }
}

// d_m = d_loss * scale * (m - 1) / norm
__global__ void __launch_bounds__(256) penalty_bwd_scaled_kernel(const float* __restrict__ m, const float* __restrict__ norm,
                                                                 const float* __restrict__ d_loss, const float scale,
                                                                 const int64_t n, float* __restrict__ d_m) {
  const float nv = norm[0];
  const float k = nv > 0.f ? d_loss[0] * scale / nv : 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    *(f4*)(d_m + 4 * i) = (*(const f4*)(m + 4 * i) - 1.0f) * k;
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t j = (n4 << 2) + threadIdx.x;
    d_m[j] = (m[j] - 1.0f) * k;
  }
}

// all masks of the model in one launch each way (blockIdx.y = mask): a launch per mask sat at the ~5 us floor
struct PenaltyMasks {
  const float* m[ACATTN_MAX_MASKS];
  float* d_m[ACATTN_MAX_MASKS];
};
__global__ void __launch_bounds__(256) penalty_partial_multi_kernel(const PenaltyMasks M, const int64_t n,
                                                                    float* __restrict__ part) {
  const float* __restrict__ m = M.m[blockIdx.y];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f4 d = 1.0f - *(const f4*)(m + 4 * i);
    acc += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
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
  if (threadIdx.x == 0) part[(size_t)blockIdx.y * kPenaltyGrid + blockIdx.x] = red[0];
}
__global__ void __launch_bounds__(256) penalty_bwd_scaled_multi_kernel(const PenaltyMasks M, const float* __restrict__ norms,
                                                                       const float* __restrict__ d_loss, const float scale,
                                                                       const int64_t n) {
  const float* __restrict__ m = M.m[blockIdx.y];
  float* __restrict__ d_m = M.d_m[blockIdx.y];
  const float nv = norms[blockIdx.y];
  const float k = nv > 0.f ? d_loss[0] * scale / nv : 0.f;
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

static int penalty_grid(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, kPenaltyGrid)); }

int acattn_launch_penalty_partial(const float* m, int64_t n, float* part, hipStream_t stream) {
  hipLaunchKernelGGL(penalty_partial_kernel, dim3(penalty_grid(n)), dim3(256), 0, stream, m, n, part);
  return (int)hipGetLastError();
}

int acattn_launch_penalty_partial_multi(const float* const* m, int n_masks, int64_t n, float* part, hipStream_t stream) {
  PenaltyMasks M{};
  for (int l = 0; l < n_masks; ++l) M.m[l] = m[l];
  hipLaunchKernelGGL(penalty_partial_multi_kernel, dim3(penalty_grid(n), n_masks), dim3(256), 0, stream, M, n, part);
  return (int)hipGetLastError();
}

int acattn_launch_penalty_bwd_scaled_multi(const float* const* m, const float* norms, const float* d_loss, float scale,
                                           int64_t n, float* const* d_m, int n_masks, hipStream_t stream) {
  PenaltyMasks M{};
  for (int l = 0; l < n_masks; ++l) M.m[l] = m[l], M.d_m[l] = d_m[l];
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, 2048));
  hipLaunchKernelGGL(penalty_bwd_scaled_multi_kernel, dim3(grid, n_masks), dim3(256), 0, stream, M, norms, d_loss, scale, n);
  return (int)hipGetLastError();
}

// pen[bh * nT + qb] = sum over rows [16 qb, 16 qb + 16) x L keys of (1 - M)^2: the block is 16 L contiguous floats
__global__ void __launch_bounds__(64) penalty_rows_kernel(const float* __restrict__ m, const int L, const int nT,
                                                          float* __restrict__ pen) {
  const int blk = blockIdx.x, bh = blk / nT, qb = blk - bh * nT;
  const int rows = min(16, L - 16 * qb), n = rows * L;
  const float* p = m + ((size_t)bh * L + 16 * qb) * L;
  float acc = 0.f;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) & 15) == 0);
  if (aligned) {
    for (int i = 4 * threadIdx.x; i + 3 < n; i += 256) {
      const f4 v = 1.0f - *(const f4*)(p + i);
      acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    for (int i = (n & ~3) + threadIdx.x; i < n; i += 64) acc += (1.0f - p[i]) * (1.0f - p[i]);
  } else {
    for (int i = threadIdx.x; i < n; i += 64) acc += (1.0f - p[i]) * (1.0f - p[i]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (threadIdx.x == 0) pen[blk] = acc;
}

struct PenaltyRows {
  const float* pen[ACATTN_MAX_MASKS];
  float* d_pen[ACATTN_MAX_MASKS];
};

// attacked_loss_finish_kernel with one pen vector per mask (see there)
__global__ void __launch_bounds__(256) attacked_loss_finish_rows_kernel(const float* __restrict__ row_loss, const int B,
                                                                        const PenaltyRows R, const int n_masks, const int count,
                                                                        const float weight, float* __restrict__ out,
                                                                        float* __restrict__ scale_buf, const int n_scale) {
  const float k = -1.0f / (float)B;
  if (blockIdx.x > 0) {
    for (int i = (blockIdx.x - 1) * 256 + threadIdx.x; i < n_scale / 4; i += (gridDim.x - 1) * 256)
      *(f4*)(scale_buf + 4 * i) = *(const f4*)(scale_buf + 4 * i) * k;
    if (blockIdx.x == 1 && threadIdx.x < (n_scale & 3)) scale_buf[(n_scale & ~3) + threadIdx.x] *= k;
    return;
  }
  __shared__ float red[256];
  auto block_sum = [&](float v) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
  };
  // one workgroup sums B + n_masks * count values (13,312 per mask at L = 200): four 16-byte loads in flight per thread --
  // a dependent dword load per 256 values was a chain of 52 L2 round trips per mask (37 us at L = 200, 11 at L = 50)
  auto thread_sum = [&](const float* __restrict__ p, const int n) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const bool vec = (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    const int n4 = vec ? n >> 2 : 0;
    const f4* p4 = (const f4*)p;
    int i = threadIdx.x;
    for (; i + 768 < n4; i += 1024) {
      const f4 v0 = p4[i], v1 = p4[i + 256], v2 = p4[i + 512], v3 = p4[i + 768];
      a0 += (v0[0] + v0[1]) + (v0[2] + v0[3]);
      a1 += (v1[0] + v1[1]) + (v1[2] + v1[3]);
      a2 += (v2[0] + v2[1]) + (v2[2] + v2[3]);
      a3 += (v3[0] + v3[1]) + (v3[2] + v3[3]);
    }
    for (; i < n4; i += 256) {
      const f4 v = p4[i];
      a0 += (v[0] + v[1]) + (v[2] + v[3]);
    }
    for (int j = 4 * n4 + threadIdx.x; j < n; j += 256) a1 += p[j];
    return (a0 + a1) + (a2 + a3);
  };
  const float ce = block_sum(thread_sum(row_loss, B)) / (float)B;
  float pen = 0.f;
  for (int l = 0; l < n_masks; ++l) {
    const float nv = sqrtf(block_sum(thread_sum(R.pen[l], count)));
    if (threadIdx.x == 0) out[2 + l] = nv;
    pen += nv;
  }
  if (threadIdx.x == 0) {
    out[0] = weight * (pen / (float)n_masks) - ce;
    out[1] = ce;
  }
}

// d_pen[l][:] = d_loss * scale / (2 norm_l); slice blockIdx.y == n_masks (when launched): d_out[:] = dir[:] * d_loss, the
// attacked loss's output cotangent from its saved direction in the same launch
__global__ void __launch_bounds__(256) penalty_drows_kernel(const PenaltyRows R, const float* __restrict__ norms,
                                                            const float* __restrict__ d_loss, const float scale, const int count,
                                                            const int n_masks, const float* __restrict__ dir,
                                                            float* __restrict__ d_out, const int n_dir) {
  if ((int)blockIdx.y >= n_masks) {  // (uniform per workgroup)
    const float k = d_loss[0];
    const int n4 = ((reinterpret_cast<uintptr_t>(dir) | reinterpret_cast<uintptr_t>(d_out)) & 15) == 0 ? n_dir >> 2 : 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) *(f4*)(d_out + 4 * i) = *(const f4*)(dir + 4 * i) * k;
    for (int i = 4 * n4 + blockIdx.x * 256 + threadIdx.x; i < n_dir; i += gridDim.x * 256) d_out[i] = dir[i] * k;
    return;
  }
  const float nv = norms[blockIdx.y];
  const float k = nv > 0.f ? d_loss[0] * scale / (2.0f * nv) : 0.f;
  float* d = R.d_pen[blockIdx.y];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) d[i] = k;
}

int acattn_launch_penalty_rows(const float* m, int B, int nh, int L, float* pen, hipStream_t stream) {
  const int nT = (L + 15) / 16;
  hipLaunchKernelGGL(penalty_rows_kernel, dim3(B * nh * nT), dim3(64), 0, stream, m, L, nT, pen);
  return (int)hipGetLastError();
}

int acattn_launch_attacked_loss_finish_rows(const float* row_loss, int B, const float* const* pen, int n_masks, int count,
                                            float weight, float* out, float* scale_buf, int n_scale, hipStream_t stream) {
  PenaltyRows R{};
  for (int l = 0; l < n_masks; ++l) R.pen[l] = pen[l];
  const int scale_wgs = n_scale > 0 ? std::min((n_scale / 4 + 255) / 256 + 1, 64) : 0;
  hipLaunchKernelGGL(attacked_loss_finish_rows_kernel, dim3(1 + scale_wgs), dim3(256), 0, stream, row_loss, B, R, n_masks,
                     count, weight, out, scale_buf, n_scale);
  return (int)hipGetLastError();
}

int acattn_launch_penalty_drows(const float* norms, const float* d_loss, float scale, int count, float* const* d_pen,
                                int n_masks, hipStream_t stream, const float* dir, float* d_out, int n_dir) {
  PenaltyRows R{};
  for (int l = 0; l < n_masks; ++l) R.d_pen[l] = d_pen[l];
  const bool with_dir = dir && d_out && n_dir > 0;
  const int gx = std::max(1, std::min((std::max(count, with_dir ? n_dir / 4 : 0) + 255) / 256, 64));
  hipLaunchKernelGGL(penalty_drows_kernel, dim3(gx, n_masks + (with_dir ? 1 : 0)), dim3(256), 0, stream, R, norms, d_loss, scale,
                     count, n_masks, dir, d_out, with_dir ? n_dir : 0);
  return (int)hipGetLastError();
}

int acattn_launch_attacked_loss_finish(const float* row_loss, int B, const float* part, int n_masks, int64_t mask_numel,
                                       float weight, float* out, float* scale_buf, int n_scale, hipStream_t stream) {
  const int scale_wgs = n_scale > 0 ? std::min((n_scale / 4 + 255) / 256 + 1, 64) : 0;
  hipLaunchKernelGGL(attacked_loss_finish_kernel, dim3(1 + scale_wgs), dim3(256), 0, stream, row_loss, B, part, n_masks,
                     penalty_grid(mask_numel), kPenaltyGrid, weight, out, scale_buf, n_scale);
  return (int)hipGetLastError();
}

int acattn_launch_penalty_bwd_scaled(const float* m, const float* norm, const float* d_loss, float scale, int64_t n,
                                     float* d_m, hipStream_t stream) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, 2048));
  hipLaunchKernelGGL(penalty_bwd_scaled_kernel, dim3(grid), dim3(256), 0, stream, m, norm, d_loss, scale, n, d_m);
  return (int)hipGetLastError();
}

int acattn_launch_penalty_bwd(const float* m, const float* norm, const float* d_norm, int64_t n, float* d_m,
                              hipStream_t stream) {
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, 2048));
  hipLaunchKernelGGL(penalty_bwd_kernel, dim3(grid), dim3(256), 0, stream, m, norm, d_norm, n, d_m);
  return (int)hipGetLastError();
}


int acattn_launch_sum_rows_pair(const float* x1, float* out1, int batch1, int R1, int C1, const float* x2, float* out2,
                                int batch2, int R2, int C2, hipStream_t stream) {
  const SumRowsJob a = make_job(x1, out1, batch1, R1, C1), b = make_job(x2, out2, batch2, R2, C2);
  hipLaunchKernelGGL(sum_rows_pair_kernel, dim3(a.n_wg + b.n_wg), dim3(256), 0, stream, a, b);
  return (int)hipGetLastError();
}

int acattn_launch_sum_rows(const float* x, float* out, int batch, int R, int C, hipStream_t stream) {
  const int c4 = (C + 3) / 4;
  const int CT = pick_ct(batch, R, C);
  const int col_groups = (c4 + CT - 1) / CT;
  // one chunk: every output element is written by exactly one workgroup (no atomics, no zero-fill).  Callers that
  // need more parallelism over a long R split it themselves into [batch * s, R / s, C] and reduce twice (ops.sum_rows)
  const dim3 grid(col_groups, 1, batch);
  hipLaunchKernelGGL(sum_rows_kernel<false>, grid, dim3(256), 0, stream, x, out, R, C, R, CT);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// zero fill as a KERNEL.  hipMemsetAsync nodes captured into a hipGraph did not reliably take effect before the kernels
// that accumulate into the buffer (float atomics into parameter partial rows / d_out): stale sums grew from replay to
// replay until the parameters went non-finite after ~70-120 steps of the L = 200, d = 128 configuration
// (tools/nan_probe_graph.py); eager launches were never affected.  Every accumulate-into-zero site uses this instead.
// ---------------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256) zero_fill_kernel(float* __restrict__ p, size_t n) {
  const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 + 3 < n && (reinterpret_cast<uintptr_t>(p + i0) & 15) == 0) {
    *(f4*)(p + i0) = f4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i0 + e < n) p[i0 + e] = 0.f;
  }
}
}  // namespace

// acattn_dense_ce_*: cross-entropy over materialised logits (include/acattn.h).  One workgroup per row in the forward (the
// row is read once: running maximum and rescaled sum per thread, folded through LDS); the backward is elementwise, four
// columns per thread.
__global__ void __launch_bounds__(256) dense_ce_fwd_kernel(const float* __restrict__ logits, long long rows, long long N,
                                                           const long long* __restrict__ target, float* __restrict__ lse,
                                                           float* __restrict__ row_loss) {
  constexpr float kL2e = 1.44269504088896340736f, kLn2 = 0.69314718055994530942f;
  __shared__ float sm[256], ss[256];
  for (long long r = blockIdx.x; r < rows; r += gridDim.x) {
    const float* x = logits + r * N;
    float m = -__builtin_inff(), s = 0.f;
    for (long long n0 = threadIdx.x; n0 < N; n0 += 1024) {  // four loads in flight per thread
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = n0 + 256 * e < N ? x[n0 + 256 * e] : -__builtin_inff();
      const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
      if (vm > m) {
        s *= __builtin_amdgcn_exp2f((m - vm) * kL2e);
        m = vm;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) s += __builtin_amdgcn_exp2f((v[e] - m) * kL2e);
    }
    sm[threadIdx.x] = m;
    ss[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
      if ((int)threadIdx.x < off) {
        const float m1 = sm[threadIdx.x], m2 = sm[threadIdx.x + off];
        const float mn = fmaxf(m1, m2);
        const float a = m1 > -__builtin_inff() ? ss[threadIdx.x] * __builtin_amdgcn_exp2f((m1 - mn) * kL2e) : 0.f;
        const float b = m2 > -__builtin_inff() ? ss[threadIdx.x + off] * __builtin_amdgcn_exp2f((m2 - mn) * kL2e) : 0.f;
        sm[threadIdx.x] = mn;
        ss[threadIdx.x] = a + b;
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const float l = sm[0] + __builtin_amdgcn_logf(ss[0]) * kLn2;
      const long long t = target[r];
      lse[r] = l;
      row_loss[r] = (t >= 0 && t < N) ? l - x[t] : __builtin_nanf("");
    }
    __syncthreads();
  }
}
__global__ void __launch_bounds__(256) dense_ce_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ lse,
                                                           const long long* __restrict__ target, const float* __restrict__ coef,
                                                           long long rows, long long N, float* __restrict__ d_logits) {
  // grid (column chunks of 1024, row groups): a thread takes columns c, c + 256, c + 512, c + 768 of its rows -- coalesced
  // dword accesses whatever the row's alignment (N = 20,001 is odd), no division per element
  constexpr float kL2e = 1.44269504088896340736f;
  const long long c0 = (long long)blockIdx.x * 1024 + threadIdx.x;
  for (long long r = blockIdx.y; r < rows; r += gridDim.y) {
    const float l = lse[r], cf = coef[r];
    const long long t = target[r];
    const float* x = logits + r * N;
    float* d = d_logits + r * N;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long n = c0 + 256 * e;
      if (n < N) d[n] = cf * (__builtin_amdgcn_exp2f((x[n] - l) * kL2e) - (n == t ? 1.0f : 0.0f));
    }
  }
}
int acattn_launch_dense_ce_fwd(const float* logits, int64_t rows, int64_t N, const int64_t* target, float* lse, float* row_loss,
                               hipStream_t stream) {
  const unsigned blocks = (unsigned)std::min<int64_t>(rows, 8192);
  hipLaunchKernelGGL(dense_ce_fwd_kernel, dim3(blocks), dim3(256), 0, stream, logits, (long long)rows, (long long)N,
                     (const long long*)target, lse, row_loss);
  return (int)hipGetLastError();
}
int acattn_launch_dense_ce_bwd(const float* logits, const float* lse, const int64_t* target, const float* coef, int64_t rows,
                               int64_t N, float* d_logits, hipStream_t stream) {
  const dim3 grid((unsigned)((N + 1023) / 1024), (unsigned)std::min<int64_t>(rows, 32768));
  hipLaunchKernelGGL(dense_ce_bwd_kernel, grid, dim3(256), 0, stream, logits, lse, (const long long*)target, coef,
                     (long long)rows, (long long)N, d_logits);
  return (int)hipGetLastError();
}

// acattn_step_inputs: the copies of a step's batch tensors, the replay counter and the read positions in one launch
// (three copyBuffer launches + two ATen elementwise launches of 4.5-5 us each per step before)
struct StepInputs {
  const unsigned char* src[ACATTN_MAX_COPIES];
  unsigned char* dst[ACATTN_MAX_COPIES];
  long long bytes[ACATTN_MAX_COPIES];
  int n;
  long long* counter;
  const long long* item_length;
  long long* last_row;
  int n_rows;
};
__global__ void __launch_bounds__(256) step_inputs_kernel(const StepInputs A) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (int k = 0; k < A.n; ++k) {
    const unsigned char* s = A.src[k];
    unsigned char* d = A.dst[k];
    const size_t nb = (size_t)A.bytes[k];
    if (((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d) | nb) & 15) == 0) {
      for (size_t i = tid; i < nb / 16; i += stride) ((uint4*)d)[i] = ((const uint4*)s)[i];
    } else {
      for (size_t i = tid; i < nb; i += stride) d[i] = s[i];
    }
  }
  if (A.last_row)
    for (size_t i = tid; i < (size_t)A.n_rows; i += stride) A.last_row[i] = A.item_length[i] - 1;
  if (A.counter && tid == 0) *A.counter += 1;
}
int acattn_launch_step_inputs(const void* const* src, void* const* dst, const int64_t* bytes, int n, int64_t* counter,
                              const int64_t* item_length, int64_t* last_row, int n_rows, hipStream_t stream) {
  StepInputs a;
  a.n = 0;
  size_t most = (size_t)(last_row ? n_rows : 0) * 16;
  for (int k = 0; k < n; ++k) {
    if (src[k] == dst[k] || bytes[k] <= 0) continue;
    a.src[a.n] = (const unsigned char*)src[k];
    a.dst[a.n] = (unsigned char*)dst[k];
    a.bytes[a.n] = bytes[k];
    most = std::max(most, (size_t)bytes[k]);
    ++a.n;
  }
  a.counter = (long long*)counter;
  a.item_length = (const long long*)item_length;
  a.last_row = (long long*)last_row;
  a.n_rows = n_rows;
  if (a.n == 0 && !counter && !last_row) return 0;
  const unsigned blocks = (unsigned)std::min<size_t>(1024, std::max<size_t>(1, (most / 16 + 255) / 256));
  hipLaunchKernelGGL(step_inputs_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return (int)hipGetLastError();
}

int acattn_launch_zero(float* p, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  // diagnosis only (tools/memset_graph_probe.py): the memset form whose captured nodes the round-2 failure involved
  static const bool use_memset = getenv("ACATTN_ZERO_MEMSET") && atoi(getenv("ACATTN_ZERO_MEMSET")) != 0;
  if (use_memset) return (int)hipMemsetAsync(p, 0, n * sizeof(float), stream);
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, p, n);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// acattn_spatial_affines: the affine planes of acattn_problem.affine from q, k and the calibrator parameters, for a
// producer that is not acattn_projections_fwd (which writes them from its accumulators).  One lane per (b, head, i):
// four dot products of length dh; padding entries [L, LP) are written as zeros.  A few MB in, ~1 MB out.
namespace {
__global__ void __launch_bounds__(256) spatial_affines_kernel(const acattn_problem P, float* affine) {
  const int L = P.L, H = P.H, nh = P.n_heads, dh = H / nh;
  const int LP = ((L + 15) >> 4) << 4;
  const long long total = (long long)P.B * nh * LP;
  const long long idx = blockIdx.x * 256LL + threadIdx.x;
  if (idx >= total) return;
  const int i = (int)(idx % LP);
  const long long bh = idx / LP;
  const int h = (int)(bh % nh);
  const long long b = bh / nh;
  float* pl = affine + bh * 4 * LP + i;
  if (i >= L) {
    pl[0] = pl[LP] = pl[2 * LP] = pl[3 * LP] = 0.f;
    return;
  }
  const float* q = P.q + ((size_t)b * L + i) * H + h * dh;
  const float* k = P.k + ((size_t)b * L + i) * H + h * dh;
  float ao = 0.f, ad = 0.f, co = 0.f, cd = 0.f;
  for (int d = 0; d < dh; d += 4) {
    const f4 qv = *(const f4*)(q + d), kv = *(const f4*)(k + d);
    const f4 wo = *(const f4*)(P.w_order + d), wd = *(const f4*)(P.w_dist + d);
    const f4 wok = *(const f4*)(P.w_order + dh + d), wdk = *(const f4*)(P.w_dist + dh + d);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ao = fmaf(qv[e], wo[e], ao);
      ad = fmaf(qv[e], wd[e], ad);
      co = fmaf(kv[e], wok[e], co);
      cd = fmaf(kv[e], wdk[e], cd);
    }
  }
  constexpr float kL2e = 1.44269504088896340736f;
  pl[0] = -kL2e * (ao + P.b_order[0]);
  pl[LP] = ad + P.b_dist[0];
  pl[2 * LP] = -kL2e * co;
  pl[3 * LP] = cd;
}
}  // namespace

int acattn_launch_spatial_affines(const acattn_problem& p, float* affine, hipStream_t stream) {
  const long long total = (long long)p.B * p.n_heads * (((p.L + 15) >> 4) << 4);
  hipLaunchKernelGGL(spatial_affines_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p, affine);
  return (int)hipGetLastError();
}

