// y = LayerNorm(dropout(z) + residual) * gamma + beta, forward and backward, one pass each, for gfx950.
//
// This is the tail of both sub-blocks of an AC-TSR encoder layer:
//   cal_adjusted_outputs   hidden = dense(ctx); hidden = out_dropout(hidden); LayerNorm(hidden + input)   layers.py:681-683
//   FeedForward.forward    hidden = dense_2(act(dense_1(x))); hidden = dropout(hidden); LayerNorm(hidden + x)  layers.py:794-796
// where torch runs dropout, add and LayerNorm (and, backward, five kernels) as separate HBM round trips.
// Pure streaming work: H (64..256) floats per row, 16-byte accesses, a row lives in H/4 adjacent lanes and its
// mean / variance are DPP reductions inside a 16-lane row (two rows' worth of permlane for H = 128, 256).
// The dropout keep decisions come from the same counter RNG family as the attention kernels (seed + optional
// device-side step counter), or from an explicit byte mask for parity tests.
#include <algorithm>

#include "acattn_common.h"
#include "acattn_rowops.h"

namespace {

__device__ __forceinline__ f4 ln_keep_scale(const acattn_ln_problem& P, uint64_t seed, int row, int c4, int H) {
  return row_keep_scale(P.p_drop, P.keep, seed, row, c4, H);
}

template <int H>
__global__ void __launch_bounds__(256) ln_fwd_kernel(const acattn_ln_problem P, float* __restrict__ y,
                                                     float* __restrict__ stats) {
  constexpr int LPR = H / 4;         // lanes per row
  constexpr int RPB = 256 / LPR;     // rows per workgroup pass
  const int c4 = threadIdx.x % LPR, rsub = threadIdx.x / LPR;
  const uint64_t seed = P.seed + (P.seed_device ? *P.seed_device : 0ull);
  const f4 gm = *(const f4*)(P.gamma + 4 * c4), bt = *(const f4*)(P.beta + 4 * c4);
  const int rows_z = P.rows;                    // z rows; the residual may have fewer (broadcast over a leading dim)
  for (int row = blockIdx.x * RPB + rsub; row < rows_z; row += gridDim.x * RPB) {
    const size_t o = (size_t)row * H + 4 * c4;
    const f4 z = *(const f4*)(P.z + o);
    const f4 r = *(const f4*)(P.residual + (size_t)(row % P.residual_rows) * H + 4 * c4);
    const f4 s = z * ln_keep_scale(P, seed, row, c4, H) + r;
    const float mean = row_sum<LPR>((s[0] + s[1]) + (s[2] + s[3])) * (1.0f / H);
    const f4 d = s - mean;
    const float var = row_sum<LPR>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / H);
    const float rstd = __builtin_amdgcn_rsqf(var + P.eps);
    *(f4*)(y + o) = (d * rstd) * gm + bt;
    if (c4 == 0) *(float2*)(stats + 2 * (size_t)row) = float2{mean, rstd};
  }
}

// dz = keep * dx, dres = dx (optionally ACCUMULATED over a broadcast leading dimension by the caller),
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  per-workgroup partials of dgamma, dbeta.
template <int H>
__global__ void __launch_bounds__(256) ln_bwd_kernel(const acattn_ln_problem P, const float* __restrict__ dy,
                                                     const float* __restrict__ stats, float* __restrict__ dz,
                                                     float* __restrict__ dres, float* __restrict__ dgb_part) {
  constexpr int LPR = H / 4;
  constexpr int RPB = 256 / LPR;
  const int c4 = threadIdx.x % LPR, rsub = threadIdx.x / LPR;
  const uint64_t seed = P.seed + (P.seed_device ? *P.seed_device : 0ull);
  const f4 gm = *(const f4*)(P.gamma + 4 * c4);
  f4 acc_g = {0.f, 0.f, 0.f, 0.f}, acc_b = acc_g;
  for (int row = blockIdx.x * RPB + rsub; row < P.rows; row += gridDim.x * RPB) {
    const size_t o = (size_t)row * H + 4 * c4;
    const f4 keep = ln_keep_scale(P, seed, row, c4, H);
    const f4 s = *(const f4*)(P.z + o) * keep + *(const f4*)(P.residual + (size_t)(row % P.residual_rows) * H + 4 * c4);
    const float2 st = *(const float2*)(stats + 2 * (size_t)row);
    const f4 xh = (s - st.x) * st.y;
    const f4 g = *(const f4*)(dy + o);
    acc_g += g * xh;
    acc_b += g;
    const f4 gg = g * gm;
    const float m1 = row_sum<LPR>((gg[0] + gg[1]) + (gg[2] + gg[3])) * (1.0f / H);
    const float m2 = row_sum<LPR>((gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3])) * (1.0f / H);
    const f4 dx = (gg - m1 - xh * m2) * st.y;
    if (dz) *(f4*)(dz + o) = dx * keep;
    if (dres) *(f4*)(dres + o) = dx;
  }
  if (dgb_part) {
    // fold the RPB row slots of the workgroup through LDS, one [2, H] partial per workgroup
    __shared__ float red[2 * 256 * 4];
    *(f4*)(red + 4 * threadIdx.x) = acc_g;
    *(f4*)(red + 1024 + 4 * threadIdx.x) = acc_b;
    __syncthreads();
    if (rsub == 0) {
      f4 sg = {0.f, 0.f, 0.f, 0.f}, sb = sg;
#pragma unroll
      for (int k = 0; k < RPB; ++k) {
        sg += *(const f4*)(red + 4 * (k * LPR + c4));
        sb += *(const f4*)(red + 1024 + 4 * (k * LPR + c4));
      }
      *(f4*)(dgb_part + (size_t)blockIdx.x * 2 * H + 4 * c4) = sg;
      *(f4*)(dgb_part + (size_t)blockIdx.x * 2 * H + H + 4 * c4) = sb;
    }
  }
}

template <int H>
int launch_fwd(const acattn_ln_problem& p, float* y, float* stats, hipStream_t stream) {
  constexpr int RPB = 256 / (H / 4);
  const int grid = (int)std::min<int64_t>(((int64_t)p.rows + RPB - 1) / RPB, 2048);
  hipLaunchKernelGGL((ln_fwd_kernel<H>), dim3(grid), dim3(256), 0, stream, p, y, stats);
  return (int)hipGetLastError();
}

template <int H>
int launch_bwd(const acattn_ln_problem& p, const float* dy, const float* stats, float* dz, float* dres, float* dgb_part,
               hipStream_t stream) {
  hipLaunchKernelGGL((ln_bwd_kernel<H>), dim3(ACATTN_LN_BWD_GRID), dim3(256), 0, stream, p, dy, stats, dz, dres, dgb_part);
  return (int)hipGetLastError();
}

}  // namespace

int acattn_launch_ln_fwd(const acattn_ln_problem& p, float* y, float* stats, hipStream_t stream) {
  switch (p.H) {
    case 64: return launch_fwd<64>(p, y, stats, stream);
    case 128: return launch_fwd<128>(p, y, stats, stream);
    case 256: return launch_fwd<256>(p, y, stats, stream);
  }
  return -1;
}

int acattn_launch_ln_bwd(const acattn_ln_problem& p, const float* dy, const float* stats, float* dz, float* dres,
                         float* dgb_part, hipStream_t stream) {
  switch (p.H) {
    case 64: return launch_bwd<64>(p, dy, stats, dz, dres, dgb_part, stream);
    case 128: return launch_bwd<128>(p, dy, stats, dz, dres, dgb_part, stream);
    case 256: return launch_bwd<256>(p, dy, stats, dz, dres, dgb_part, stream);
  }
  return -1;
}
