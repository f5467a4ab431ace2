// y = dropout(LayerNorm(item_embedding[idx] + position_embedding)) and its backward, one launch each way, gfx950.
//
// The front end of both models (recbole/model/sequential_recommender/acsasrec.py:87-95, acbert4rec.py:163-171):
// torch runs gather, add, LayerNorm and dropout as four HBM round trips forward and, backward, a LayerNorm
// backward (3 kernels), a dropout scale and an index_add over a zeroed [N, H] table gradient.  Here:
//   forward  : a row (H/4 adjacent lanes, 16 bytes each) is gathered straight from the table, normalised and
//              written once; (mean, 1/std) are kept for the backward;
//   backward : workgroup (l, chunk) walks the sequences of its chunk at ONE position l, so the position-embedding
//              gradient accumulates in registers; the row is re-gathered (cheaper than storing x-hat), the
//              LayerNorm backward is formed in registers and scattered into the table gradient with float
//              atomics -- rows of `padding_idx` are skipped, as nn.Embedding does (and because half of all lookups
//              hit that one row: thousands of atomics on 256 bytes serialise).
#include <algorithm>

#include "acattn_common.h"
#include "acattn_rowops.h"

namespace {

__device__ __forceinline__ int64_t safe_id(const int64_t* idx, int row, int64_t n) {
  const int64_t id = idx[row];
  return id < 0 ? 0 : (id >= n ? n - 1 : id);  // an out-of-range id must not become an out-of-bounds read
}

template <int H>
__global__ void __launch_bounds__(256) embed_ln_fwd_kernel(const acattn_embed_problem P, float* __restrict__ y,
                                                           float* __restrict__ stats) {
  constexpr int LPR = H / 4;
  constexpr int RPB = 256 / LPR;
  const int c4 = threadIdx.x % LPR, rsub = threadIdx.x / LPR;
  const uint64_t seed = P.seed + (P.seed_device ? *P.seed_device : 0ull);
  const f4 gm = *(const f4*)(P.gamma + 4 * c4), bt = *(const f4*)(P.beta + 4 * c4);
  for (int row = blockIdx.x * RPB + rsub; row < P.rows; row += gridDim.x * RPB) {
    const int64_t id = safe_id(P.idx, row, P.n_table_rows);
    f4 s = *(const f4*)(P.table + id * H + 4 * c4);
    if (P.pos) s += *(const f4*)(P.pos + (size_t)(row % P.L) * H + 4 * c4);
    const float mean = row_sum<LPR>((s[0] + s[1]) + (s[2] + s[3])) * (1.0f / H);
    const f4 d = s - mean;
    const float var = row_sum<LPR>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / H);
    const float rstd = __builtin_amdgcn_rsqf(var + P.eps);
    const f4 keep = row_keep_scale(P.p_drop, P.keep, seed, row, c4, H);
    *(f4*)(y + (size_t)row * H + 4 * c4) = ((d * rstd) * gm + bt) * keep;
    if (c4 == 0) {
      *(float2*)(stats + 2 * (size_t)row) = float2{mean, rstd};
      if (P.nonzero_out) P.nonzero_out[row] = P.idx[row] != 0;
    }
  }
}

template <int H>
__global__ void __launch_bounds__(256) embed_ln_bwd_kernel(const acattn_embed_problem P, const float* __restrict__ dy,
                                                           const float* __restrict__ stats, const int64_t padding_idx,
                                                           float* __restrict__ d_table, float* __restrict__ d_pos_part,
                                                           float* __restrict__ dgb_part) {
  constexpr int LPR = H / 4;
  constexpr int RPB = 256 / LPR;
  const int c4 = threadIdx.x % LPR, rsub = threadIdx.x / LPR;
  const int l = blockIdx.x, bc = blockIdx.y, BC = gridDim.y;
  const int B = P.rows / P.L;
  const int per = (B + BC - 1) / BC;
  const int b_end = min(B, (bc + 1) * per);
  const uint64_t seed = P.seed + (P.seed_device ? *P.seed_device : 0ull);
  const f4 gm = *(const f4*)(P.gamma + 4 * c4);
  f4 pos = {0.f, 0.f, 0.f, 0.f};
  if (P.pos) pos = *(const f4*)(P.pos + (size_t)l * H + 4 * c4);
  f4 acc_g = {0.f, 0.f, 0.f, 0.f}, acc_b = acc_g, acc_p = acc_g, acc_hot = acc_g;
  const int64_t hot = P.hot_id_plus1 - 1;  // -1: none
  for (int b = bc * per + rsub; b < b_end; b += RPB) {
    const int row = b * P.L + l;
    const int64_t id = safe_id(P.idx, row, P.n_table_rows);
    const size_t o = (size_t)row * H + 4 * c4;
    const f4 s = *(const f4*)(P.table + id * H + 4 * c4) + pos;
    const float2 st = *(const float2*)(stats + 2 * (size_t)row);
    const f4 xh = (s - st.x) * st.y;
    const f4 g = *(const f4*)(dy + o) * row_keep_scale(P.p_drop, P.keep, seed, row, c4, H);  // dropout sits AFTER the norm
    acc_g += g * xh;
    acc_b += g;
    const f4 gg = g * gm;
    const float m1 = row_sum<LPR>((gg[0] + gg[1]) + (gg[2] + gg[3])) * (1.0f / H);
    const float m2 = row_sum<LPR>((gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3])) * (1.0f / H);
    const f4 dx = (gg - m1 - xh * m2) * st.y;
    acc_p += dx;
    if (d_table && id != padding_idx) {
      if (id == hot) {  // [r4] the one row a fifth of all lookups hit: summed in the workgroup, one atomic per column below
        acc_hot += dx;
      } else {
        float* t = d_table + id * H + 4 * c4;
#pragma unroll
        for (int e = 0; e < 4; ++e) unsafeAtomicAdd(t + e, dx[e]);
      }
    }
  }
  // fold the RPB row slots of the workgroup through LDS
  __shared__ float red[3 * 256 * 4];
  *(f4*)(red + 4 * threadIdx.x) = acc_g;
  *(f4*)(red + 1024 + 4 * threadIdx.x) = acc_b;
  *(f4*)(red + 2048 + 4 * threadIdx.x) = acc_p;
  __syncthreads();
  if (rsub == 0) {
    f4 sg = {0.f, 0.f, 0.f, 0.f}, sb = sg, sp = sg;
#pragma unroll
    for (int k = 0; k < RPB; ++k) {
      sg += *(const f4*)(red + 4 * (k * LPR + c4));
      sb += *(const f4*)(red + 1024 + 4 * (k * LPR + c4));
      sp += *(const f4*)(red + 2048 + 4 * (k * LPR + c4));
    }
    const size_t wg = (size_t)bc * P.L + l;
    if (dgb_part) {
      *(f4*)(dgb_part + wg * 2 * H + 4 * c4) = sg;
      *(f4*)(dgb_part + wg * 2 * H + H + 4 * c4) = sb;
    }
    if (d_pos_part) *(f4*)(d_pos_part + wg * H + 4 * c4) = sp;  // [BC, L, H]
  }
  if (hot >= 0 && d_table && hot != padding_idx) {  // (wave-uniform)
    __syncthreads();
    *(f4*)(red + 4 * threadIdx.x) = acc_hot;
    __syncthreads();
    if (rsub == 0) {
      f4 sh = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < RPB; ++k) sh += *(const f4*)(red + 4 * (k * LPR + c4));
      float* t = d_table + hot * H + 4 * c4;
#pragma unroll
      for (int e = 0; e < 4; ++e) unsafeAtomicAdd(t + e, sh[e]);
    }
  }
}

template <int H>
int launch_fwd(const acattn_embed_problem& p, float* y, float* stats, hipStream_t stream) {
  constexpr int RPB = 256 / (H / 4);
  const int grid = (int)std::min<int64_t>(((int64_t)p.rows + RPB - 1) / RPB, 2048);
  hipLaunchKernelGGL((embed_ln_fwd_kernel<H>), dim3(grid), dim3(256), 0, stream, p, y, stats);
  return (int)hipGetLastError();
}

template <int H>
int launch_bwd(const acattn_embed_problem& p, const float* dy, const float* stats, int64_t padding_idx, float* d_table,
               float* d_pos_part, float* dgb_part, hipStream_t stream) {
  hipLaunchKernelGGL((embed_ln_bwd_kernel<H>), dim3(p.L, ACATTN_EMBED_BWD_CHUNKS), dim3(256), 0, stream, p, dy, stats,
                     padding_idx, d_table, d_pos_part, dgb_part);
  return (int)hipGetLastError();
}

}  // namespace

int acattn_launch_embed_fwd(const acattn_embed_problem& p, float* y, float* stats, hipStream_t stream) {
  switch (p.H) {
    case 64: return launch_fwd<64>(p, y, stats, stream);
    case 128: return launch_fwd<128>(p, y, stats, stream);
    case 256: return launch_fwd<256>(p, y, stats, stream);
  }
  return -1;
}

int acattn_launch_embed_bwd(const acattn_embed_problem& p, const float* dy, const float* stats, int64_t padding_idx,
                            float* d_table, float* d_pos_part, float* dgb_part, hipStream_t stream) {
  switch (p.H) {
    case 64: return launch_bwd<64>(p, dy, stats, padding_idx, d_table, d_pos_part, dgb_part, stream);
    case 128: return launch_bwd<128>(p, dy, stats, padding_idx, d_table, d_pos_part, dgb_part, stream);
    case 256: return launch_bwd<256>(p, dy, stats, padding_idx, d_table, d_pos_part, dgb_part, stream);
  }
  return -1;
}
