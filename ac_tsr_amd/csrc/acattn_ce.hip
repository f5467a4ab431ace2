// Full-catalogue cross-entropy without materialising the [B, N] logits, for gfx950 (MI355X).
//
// Replaces ACSASRec._cal_loss for loss_type 'CE' (recbole/model/sequential_recommender/acsasrec.py:117-120):
//     logits = output @ item_embedding.weight^T          [B, N]   (205 MB at 512 x 100k, written and re-read 3x)
//     loss   = CrossEntropyLoss(logits, pos_items)
// and its backward.  It is attention with Q = output, K = V = the item table and one "head":
//     row_loss[b] = logsumexp_n(out_b . E_n) - out_b . E_target(b)
//     d out_b     = coef_b * (sum_n p_bn E_n - E_target(b)),      d E_n = sum_b coef_b (p_bn - [n == target_b]) out_b
// Items are stationary: a wave owns 16*TILES table rows for the whole kernel and sweeps the batch in 16-row
// blocks, computing logits^T = E . out^T with exact-fp32 MFMA (16x16x4).  In that orientation the batch row sits
// on the lane, so the row statistics are register-local, the logit registers are directly the B operand of
// d out^T = E^T . dl^T, and d E accumulates in registers over the sweep (no atomics anywhere).  What crosses
// workgroups is small: (max, sum-exp) pairs per (row, wave) in the forward, one [B, H] slab of d out per
// workgroup in the backward, each folded by a second tiny kernel.
#include <algorithm>

#include <stdlib.h>
#include <string.h>

#include "acattn_common.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr int CE_NW = 4;  // waves per workgroup

// NTILES (16-item tiles per wave) is picked per launch so that one round of workgroups covers the catalogue
// (e.g. 100k items: 7 tiles -> 224 workgroups of 448 items on 256 CUs) instead of a ragged second round.
template <int CH, int NTILES>
struct CeCfg {
  static constexpr int KS = CH / 4;         // k-steps of a logit tile == fragment floats per lane
  static constexpr int DT = CH / 16;        // 16-wide column tiles of the hidden dimension
  static constexpr int TILES = NTILES;      // 16-item tiles per wave
  static constexpr int ITEMS = 16 * TILES;  // table rows owned by one wave
  static constexpr int ES = CH + 4;         // padded LDS row stride
};

// ---------------------------------------------------------------------------------------------------------
// forward: per (row, wave) partial (max, sum exp) of the wave's items
// ---------------------------------------------------------------------------------------------------------
// SPLIT [round 3]: the grid covers gridDim.x * CE_NW * ITEMS items (a whole number of NTILES-tile waves, fewer than N) and
// the n_left 16-item tiles behind them are LEFTOVER tiles: workgroup l < n_left also takes leftover tile l, its four
// waves a quarter of the batch sweep each, after their own sweep.  100,000 items are 6,250 tiles = 1,024 waves x 6 + 106:
// with seven tiles per wave (224 workgroups) every launch lasts seven tiles' sweep, with six + a quarter sweep of one
// more tile for 106 workgroups 6.25.  The leftover tile's results are separate partials / slabs behind the regular ones
// (one writer per row: the waves' row blocks are disjoint), so nothing is read back or accumulated in place.
template <int CH, int NTILES, bool SPLIT = false>
__global__ void __launch_bounds__(64 * CE_NW) ce_fwd_kernel(const acattn_ce_problem P, float2* __restrict__ part, int n_left = 0) {
  using C = CeCfg<CH, NTILES>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int wid = blockIdx.x * CE_NW + wave;
  const int item0 = wid * C::ITEMS;
  const int B = P.B, N = P.N;

  float ef[C::TILES][C::KS];  // A operand: table rows of this wave, lane (c, g) holds row 16t+c, columns KS*g ..
#pragma unroll
  for (int t = 0; t < C::TILES; ++t) {
    const int item = item0 + 16 * t + c;
#pragma unroll
    for (int s4 = 0; s4 < C::KS / 4; ++s4) {
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (item < N) v = *(const f4*)(P.table + (size_t)item * CH + C::KS * g + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) ef[t][4 * s4 + e] = v[e];
    }
  }
  const int nrb = (B + 15) >> 4;
  const bool ragged = item0 + C::ITEMS > N;  // (uniform per wave)
  float hf[C::KS];
  auto load_rows = [&](int rb, float (&dst)[C::KS]) {
    const int row = 16 * rb + c;
#pragma unroll
    for (int s4 = 0; s4 < C::KS / 4; ++s4) {
      f4 v = {0.f, 0.f, 0.f, 0.f};
      if (row < B) v = *(const f4*)(P.out + (size_t)row * CH + C::KS * g + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[4 * s4 + e] = v[e];
    }
  };
  load_rows(0, hf);
  for (int rb = 0; rb < nrb; ++rb) {
    float hn[C::KS];
    if (rb + 1 < nrb) load_rows(rb + 1, hn);  // next block's rows are in flight under this block's MFMAs
    f4 acc[C::TILES];
#pragma unroll
    for (int t = 0; t < C::TILES; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    // k-step outer, tiles inner: consecutive MFMAs hit different accumulators (a dependent 16x16x4 chain
    // issues every 40 cycles, independent ones every 32)
#pragma unroll
    for (int s = 0; s < C::KS; ++s)
#pragma unroll
      for (int t = 0; t < C::TILES; ++t) acc[t] = mfma16(ef[t][s], hf[s], acc[t]);
    // The fp32 MFMAs do not run beside VALU work on this chip (tools/probe/coexec.hip), so every instruction between two
    // row blocks' products counts: the catalogue-end test only in the one wave that straddles the end (uniform), the
    // exponent as one fused multiply-add on register pairs (v_pk_fma_f32), the sum on pairs as well.
    if (ragged) {
#pragma unroll
      for (int t = 0; t < C::TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (item0 + 16 * t + 4 * g + r >= N) acc[t][r] = ACATTN_NEG_INF;
    }
    float m = ACATTN_NEG_INF;
#pragma unroll
    for (int t = 0; t < C::TILES; ++t) m = fmaxf(fmaxf(fmaxf(fmaxf(m, acc[t][0]), acc[t][1]), acc[t][2]), acc[t][3]);  // 2 x v_max3
    m = quad_max(m);
    float sum = 0.f;
    if (m > ACATTN_NEG_INF) {
      const float m2 = m * kLog2e;
      f4 sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < C::TILES; ++t) {
        f4 x = acc[t] * kLog2e - m2;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
        sv += x;
      }
      sum = (sv[0] + sv[1]) + (sv[2] + sv[3]);
    }
    sum = quad_sum(sum);
    const int row = 16 * rb + c;
    if (g == 0 && row < B) part[(size_t)wid * B + row] = float2{m, sum};
    if (rb + 1 < nrb) {
#pragma unroll
      for (int s = 0; s < C::KS; ++s) hf[s] = hn[s];
    }
  }
  if constexpr (SPLIT) {
    // leftover units (tile, row block), n_left * nrb of them, dealt out evenly: wave w takes units [w U, (w + 1) U)
    const int n_units = n_left * nrb, U = (n_units + gridDim.x * CE_NW - 1) / (gridDim.x * CE_NW);
    const int first = wid * U, n_my = min(max(n_units - first, 0), U);
    int cur_tile = -1;
    float ex[C::KS];
    auto load_tile = [&](int tile) {
      const int itx = gridDim.x * CE_NW * C::ITEMS + 16 * tile;
#pragma unroll
      for (int s4 = 0; s4 < C::KS / 4; ++s4) {
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (itx + c < N) v = *(const f4*)(P.table + (size_t)(itx + c) * CH + C::KS * g + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) ex[4 * s4 + e] = v[e];
      }
    };
    if (n_my > 0) {  // (requested together with the first units' rows, not after them)
      cur_tile = first / nrb;
      load_tile(cur_tile);
    }
    constexpr int GU = 4;  // units whose batch rows are requested together: a unit is too short to cover the next one's round trip
    for (int k0 = 0; k0 < n_my; k0 += GU) {
      float hx[GU][C::KS];
#pragma unroll
      for (int q = 0; q < GU; ++q)
        if (k0 + q < n_my) load_rows((first + k0 + q) % nrb, hx[q]);
#pragma unroll
      for (int q = 0; q < GU; ++q) {
        if (k0 + q >= n_my) break;
        const int u = first + k0 + q, tile = u / nrb, rb = u - tile * nrb;
        const int itx = gridDim.x * CE_NW * C::ITEMS + 16 * tile;  // first item of the leftover tile
        if (tile != cur_tile) {  // (uniform per wave)
          cur_tile = tile;
          load_tile(tile);
        }
        f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
        for (int s = 0; s < C::KS; s += 2) {
          a0 = mfma16(ex[s], hx[q][s], a0);
          a1 = mfma16(ex[s + 1], hx[q][s + 1], a1);
        }
        f4 acc = a0 + a1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (itx + 4 * g + r >= N) acc[r] = ACATTN_NEG_INF;
        const float m = quad_max(fmaxf(fmaxf(acc[0], acc[1]), fmaxf(acc[2], acc[3])));
        float sum = 0.f;
        if (m > ACATTN_NEG_INF) {
          f4 x = acc * kLog2e - m * kLog2e;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
          sum = (x[0] + x[1]) + (x[2] + x[3]);
        }
        sum = quad_sum(sum);
        const int row = 16 * rb + c;
        if (g == 0 && row < B) part[(size_t)(gridDim.x * CE_NW + tile) * B + row] = float2{m, sum};
      }
    }
  }
}

// one wave per batch row: fold the partials, add the target logit
template <int CH>
__global__ void __launch_bounds__(256) ce_fwd_reduce_kernel(const acattn_ce_problem P, const float2* __restrict__ part,
                                                            int n_part, float* __restrict__ lse,
                                                            float* __restrict__ row_loss) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= P.B) return;
  float m = ACATTN_NEG_INF, s = 0.f;
  for (int k = lane; k < n_part; k += 64) {
    const float2 p = part[(size_t)k * P.B + row];
    if (p.x > m) {
      s = s * __builtin_amdgcn_exp2f((m - p.x) * kLog2e) + p.y;
      m = p.x;
    } else if (p.x > ACATTN_NEG_INF) {
      s += p.y * __builtin_amdgcn_exp2f((p.x - m) * kLog2e);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float m2 = __shfl_xor(m, off), s2 = __shfl_xor(s, off);
    const float mn = fmaxf(m, m2);
    const float a = m > ACATTN_NEG_INF ? s * __builtin_amdgcn_exp2f((m - mn) * kLog2e) : 0.f;
    const float b = m2 > ACATTN_NEG_INF ? s2 * __builtin_amdgcn_exp2f((m2 - mn) * kLog2e) : 0.f;
    m = mn;
    s = a + b;
  }
  // A target outside [0, N) (an ignore_index such as -100, a corrupt label) must not become an out-of-bounds read of
  // the table: the row is clamped for the address and its loss is NaN, which the trainer's NaN guard reports
  // (torch's CrossEntropyLoss raises a device assert in this case).
  const long long tgt_raw = P.target[row];
  const bool tgt_ok = tgt_raw >= 0 && tgt_raw < P.N;
  const long long tgt = tgt_ok ? tgt_raw : 0;
  float dot = 0.f;
  for (int d = lane; d < CH; d += 64) dot += P.out[(size_t)row * CH + d] * P.table[(size_t)tgt * CH + d];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
  if (lane == 0) {
    const float l = m + __builtin_amdgcn_logf(s) * kLn2;
    lse[row] = l;
    row_loss[row] = tgt_ok ? l - dot : __builtin_nanf("");
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------
// Diagnostic builds only (-DACATTN_CE_STAMPS, tools/gpu_ce_stamps.sh): cycles per phase of the row-block loop, summed
// over the row blocks of a wave in scalar registers and written once at the end to g_ce_stamps[wave][8].
#ifdef ACATTN_CE_STAMPS
__device__ unsigned long long g_ce_stamps[4096 * 8];
#define CE_STAMP(k)                                                                \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    unsigned long long now_;                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");   \
    if ((k) > 0) cyc_[(k) > 0 ? (k) - 1 : 0] += now_ - last_;                      \
    last_ = now_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define CE_STAMP(k)
#endif

// DIR = true turns the same sweep into a FORWARD that also yields the direction of d_out: lse is not known yet, so
// every wave exponentiates against its own running maximum, the four waves of a workgroup are brought to a common
// maximum when their [16, CH] tiles are folded, and the workgroup emits (max, sum-exp) per row next to its slab of
// sum_n exp(l_n - max) E_n; ce_dir_reduce_kernel finishes the soft-max across workgroups (flash-attention with
// K = V = the item table).  `lse` / `coef` are unused then, `part` receives the (max, sum-exp) pairs.
template <int CH, int NTILES, bool WITH_TABLE_GRAD, bool DIR = false, bool SPLIT = false>
__global__ void __launch_bounds__(64 * CE_NW) ce_bwd_kernel(const acattn_ce_problem P, const float* __restrict__ lse,
                                                            const float* __restrict__ coef,
                                                            float* __restrict__ d_out,
                                                            float* __restrict__ d_out_slab,
                                                            float* __restrict__ d_table,
                                                            float2* __restrict__ part = nullptr, int n_left = 0) {
  static_assert(!(DIR && WITH_TABLE_GRAD), "the forward-with-direction sweep has no table gradient");
  using C = CeCfg<CH, NTILES>;
  constexpr int TS = C::ITEMS + 16 + ((C::ITEMS / 16 + 1) % 2 ? 0 : 16);  // transpose-scratch row stride: 16 * odd -> conflict-free column reads
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int item0 = (blockIdx.x * CE_NW + wave) * C::ITEMS;
  const bool ragged = item0 + C::ITEMS > P.N;  // (uniform per wave)
  const int B = P.B, N = P.N;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Es = smem + wave * C::ITEMS * C::ES;               // [ITEMS][ES]  this wave's table rows
  float* Hs = smem + CE_NW * C::ITEMS * C::ES;              // [16][ES]     current batch rows (shared)
  float* X = Hs + 16 * C::ES;                               // per-wave exchange area, max(16*TS, 16*ES) floats
  constexpr int XS = (16 * TS > 16 * C::ES) ? 16 * TS : 16 * C::ES;
  float* Xw = X + wave * XS;

  // stage this wave's table rows (zero past the catalogue end)
  for (int idx = lane; idx < C::ITEMS * (CH / 4); idx += 64) {
    const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (item0 + r < N) v = *(const f4*)(P.table + (size_t)(item0 + r) * CH + 4 * c4);
    *(f4*)(Es + r * C::ES + 4 * c4) = v;
  }
  f4 dE[C::TILES][C::DT];
#pragma unroll
  for (int t = 0; t < C::TILES; ++t)
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) dE[t][dt] = f4{0.f, 0.f, 0.f, 0.f};

  const int nrb = (B + 15) >> 4;
  // The batch rows of block rb+1 (and their lse / coef / target) are fetched into registers while block rb is on the
  // matrix cores: with one wave per SIMD nothing else would hide that L2 round trip, 32+ times per launch.
  constexpr int HV = (16 * (CH / 4) + 64 * CE_NW - 1) / (64 * CE_NW);  // float4s of a [16, CH] block per thread
  f4 h_next[HV];
  float lse_next = 0.f, cf_next = 0.f;
  int tgt_next = -1;
  auto prefetch = [&](int rb) {
#pragma unroll
    for (int u = 0; u < HV; ++u) {
      const int idx = threadIdx.x + u * 64 * CE_NW;
      const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
      h_next[u] = f4{0.f, 0.f, 0.f, 0.f};
      if (idx < 16 * (CH / 4) && 16 * rb + r < B) h_next[u] = *(const f4*)(P.out + (size_t)(16 * rb + r) * CH + 4 * c4);
    }
    const int row = 16 * rb + c;
    const bool ok = row < B;
    lse_next = (ok && !DIR) ? lse[row] : 0.f;
    cf_next = (ok && !DIR) ? coef[P.coef_is_scalar ? 0 : row] * (P.coef_scale != 0.f ? P.coef_scale : 1.0f) : 0.f;
    tgt_next = ok ? (int)P.target[row] : -1;
  };
  prefetch(0);
#ifdef ACATTN_CE_STAMPS
  unsigned long long cyc_[8] = {}, last_ = 0;
#endif
  for (int rb = 0; rb < nrb; ++rb) {
    CE_STAMP(0);
    __syncthreads();  // Hs and the exchange area of the previous block are free
#pragma unroll
    for (int u = 0; u < HV; ++u) {
      const int idx = threadIdx.x + u * 64 * CE_NW;
      const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
      if (idx < 16 * (CH / 4)) *(f4*)(Hs + r * C::ES + 4 * c4) = h_next[u];
    }
    const bool row_ok = 16 * rb + c < B;
    const float l2 = lse_next * kLog2e;
    const float cf = cf_next;
    const int tgt = row_ok ? tgt_next - item0 : -1;  // target as an index into this wave's items
    if (rb + 1 < nrb) prefetch(rb + 1);
    __syncthreads();

    CE_STAMP(1);
    // logits^T tile set and dl = coef * (softmax - onehot), layout: lane = batch row, registers = items
    float hf[C::KS];
#pragma unroll
    for (int s4 = 0; s4 < C::KS / 4; ++s4) {
      const f4 v = *(const f4*)(Hs + c * C::ES + C::KS * g + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) hf[4 * s4 + e] = v[e];
    }
    // The LDS reads of an operand group are issued TWO groups ahead of the MFMAs that use them, into their own
    // registers: with one wave per SIMD nothing else hides the LDS latency, and left to itself the compiler reuses one
    // register set for consecutive groups, i.e. read -> wait -> 4 MFMAs -> read ... (a third of this loop was waiting).
    f4 dl[C::TILES];
    {
      constexpr int S4 = C::KS / 4, NG = C::TILES * S4;  // operand groups: 4 k-steps of one item tile
      auto e_group = [&](int k) -> f4 { return *(const f4*)(Es + (16 * (k / S4) + c) * C::ES + C::KS * g + 4 * (k % S4)); };
      f4 eb[3];
      eb[0] = e_group(0);
      eb[1] = e_group(NG > 1 ? 1 : 0);
      f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;  // two interleaved chains (see the forward)
#pragma unroll
      for (int k = 0; k < NG; ++k) {
        if (k + 2 < NG) eb[(k + 2) % 3] = e_group(k + 2);
        __builtin_amdgcn_sched_barrier(0);  // the read stays above this group's MFMAs (the scheduler would sink it to its use)
        const f4 e4 = eb[k % 3];
        const int s4 = k % S4;
        a0 = mfma16(e4[0], hf[4 * s4 + 0], a0);
        a1 = mfma16(e4[1], hf[4 * s4 + 1], a1);
        a0 = mfma16(e4[2], hf[4 * s4 + 2], a0);
        a1 = mfma16(e4[3], hf[4 * s4 + 3], a1);
        if (s4 == S4 - 1) {
          dl[k / S4] = a0 + a1;
          a0 = f4{0.f, 0.f, 0.f, 0.f};
          a1 = a0;
        }
      }
    }
    CE_STAMP(2);
    // (as in the forward: the catalogue-end test only in the wave that straddles the end, the exponents' arguments and the
    // scaling on register pairs -- these instructions do not run beside the fp32 MFMAs)
    if (ragged) {
#pragma unroll
      for (int t = 0; t < C::TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (item0 + 16 * t + 4 * g + r >= N) dl[t][r] = ACATTN_NEG_INF;  // exp2(-inf) = 0 past the catalogue end
    }
    float m_w = ACATTN_NEG_INF, s_w = 0.f;  // DIR: this wave's maximum and sum-exp for batch row c
    if (DIR) {
#pragma unroll
      for (int t = 0; t < C::TILES; ++t) m_w = fmaxf(fmaxf(fmaxf(fmaxf(m_w, dl[t][0]), dl[t][1]), dl[t][2]), dl[t][3]);
      m_w = quad_max(m_w);
      const float m2 = m_w > ACATTN_NEG_INF ? m_w * kLog2e : 0.f;
      f4 sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < C::TILES; ++t) {
        f4 x = dl[t] * kLog2e - m2;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
        dl[t] = x;
        sv += x;
      }
      s_w = quad_sum((sv[0] + sv[1]) + (sv[2] + sv[3]));
    } else {
      // the target's one-hot: the tile and register that hold it are found once per row block, not tested per element
      const int t_t = tgt >> 4, g_t = (tgt >> 2) & 3, r_t = tgt & 3;  // (tgt < 0: no tile matches)
      const bool mine = tgt >= 0 && tgt < C::ITEMS && g_t == g;
#pragma unroll
      for (int t = 0; t < C::TILES; ++t) {
        f4 x = dl[t] * kLog2e - l2;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
        if (mine && t == t_t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] -= (r == r_t) ? 1.0f : 0.0f;
        }
        dl[t] = x * cf;
      }
    }
    CE_STAMP(3);
    // d out^T (this wave's items) = E^T . dl^T : dl registers are the B operand as they stand
    f4 dh[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) dh[dt] = f4{0.f, 0.f, 0.f, 0.f};
    {
      constexpr int NR = C::TILES * 4;  // item rows of this lane group, one k-step each; E^T values two rows ahead
      float et[3][C::DT];
      auto e_row = [&](int k, float (&dst)[C::DT]) {
        const float* ep = Es + (16 * (k >> 2) + 4 * g + (k & 3)) * C::ES + c;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) dst[dt] = ep[16 * dt];
      };
      e_row(0, et[0]);
      e_row(1, et[1]);
#pragma unroll
      for (int k = 0; k < NR; ++k) {
        if (k + 2 < NR) e_row(k + 2, et[(k + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) dh[dt] = mfma16(et[k % 3][dt], dl[k >> 2][k & 3], dh[dt]);
      }
    }
    CE_STAMP(4);
    if (WITH_TABLE_GRAD) {
      // d E (this wave's items) += dl^T . out : transpose dl through the wave's exchange area
#pragma unroll
      for (int t = 0; t < C::TILES; ++t) *(f4*)(Xw + c * TS + 16 * t + 4 * g) = dl[t];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float bv[4][C::DT];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) bv[s][dt] = Hs[(4 * s + g) * C::ES + 16 * dt + c];
      float at[2][4];  // transposed dl of an item tile, read one tile ahead
#pragma unroll
      for (int s = 0; s < 4; ++s) at[0][s] = Xw[(4 * s + g) * TS + c];
#pragma unroll
      for (int t = 0; t < C::TILES; ++t) {
        if (t + 1 < C::TILES) {
#pragma unroll
          for (int s = 0; s < 4; ++s) at[(t + 1) & 1][s] = Xw[(4 * s + g) * TS + 16 * (t + 1) + c];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) dE[t][dt] = mfma16(at[t & 1][s], bv[s][dt], dE[t][dt]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    CE_STAMP(5);
    // fold the four waves' d out tiles: each wave parks its [16][CH] tile, then every thread sums one float4
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) *(f4*)(Xw + c * C::ES + 16 * dt + 4 * g) = dh[dt];
    if (DIR && g == 0) {  // (max, sum-exp) of this wave for row c, in the pad columns of its parked tile
      Xw[c * C::ES + CH] = m_w;
      Xw[c * C::ES + CH + 1] = s_w;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 16 * (CH / 4); idx += blockDim.x) {
      const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
      if (16 * rb + r < B) {
        f4 sum = {0.f, 0.f, 0.f, 0.f};
        const size_t o = (size_t)(16 * rb + r) * CH + 4 * c4;
        if (DIR) {
          float mw[CE_NW], m_wg = ACATTN_NEG_INF;
#pragma unroll
          for (int w = 0; w < CE_NW; ++w) {
            mw[w] = X[w * XS + r * C::ES + CH];
            m_wg = fmaxf(m_wg, mw[w]);
          }
          float s_wg = 0.f;
#pragma unroll
          for (int w = 0; w < CE_NW; ++w) {
            const float sc = mw[w] > ACATTN_NEG_INF ? __builtin_amdgcn_exp2f((mw[w] - m_wg) * kLog2e) : 0.f;
            sum += *(const f4*)(X + w * XS + r * C::ES + 4 * c4) * sc;
            s_wg += X[w * XS + r * C::ES + CH + 1] * sc;
          }
          *(f4*)(d_out_slab + (size_t)blockIdx.x * B * CH + o) = sum;
          if (c4 == 0) part[(size_t)blockIdx.x * B + 16 * rb + r] = float2{m_wg, s_wg};
          continue;
        }
#pragma unroll
        for (int w = 0; w < CE_NW; ++w) sum += *(const f4*)(X + w * XS + r * C::ES + 4 * c4);
        // one [16, CH] tile per workgroup and row block.  Normally it goes to the workgroup's own [B, CH] slab and
        // ce_bwd_reduce_kernel folds the slabs (plain stores: 7.3 M float atomics per call cost 64 us of 330).
        // With very many rows (slabs beyond kSlabLimit) it is added to d_out with float atomics instead.
        if (d_out_slab) {
          *(f4*)(d_out_slab + (size_t)blockIdx.x * B * CH + o) = sum;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) atomicAdd(d_out + o + e, sum[e]);
        }
      }
    }
    CE_STAMP(6);
  }
#ifdef ACATTN_CE_STAMPS
  if (lane == 0 && blockIdx.x * CE_NW + wave < 4096)
    for (int k = 0; k < 8; ++k) g_ce_stamps[(blockIdx.x * CE_NW + wave) * 8 + k] = cyc_[k];
#endif
  if (WITH_TABLE_GRAD) {
#pragma unroll
    for (int t = 0; t < C::TILES; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int item = item0 + 16 * t + 4 * g + r;
        if (item < N) {
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) d_table[(size_t)item * CH + 16 * dt + c] = dE[t][dt][r];
        }
      }
  }
  if constexpr (SPLIT) {
    // Leftover units (tile, row block), see ce_fwd_kernel.  Without a table gradient they are dealt out evenly over all
    // waves; with one, workgroup l < n_left takes ALL of tile l (its waves every fourth row block), so that the tile's
    // d_table rows are summed inside the workgroup.  Every wave works on its own: its table-row area in LDS is free now and
    // holds the tile's rows, the wave's batch rows and its transpose scratch; no workgroup barrier inside the sweep.
    // Results: slab / partial number gridDim.x + tile.
    const int wid = blockIdx.x * CE_NW + wave;
    int first, stride, n_my;
    if (WITH_TABLE_GRAD) {
      first = blockIdx.x * nrb + wave;
      stride = CE_NW;
      n_my = (int)blockIdx.x < n_left ? (nrb - wave + CE_NW - 1) / CE_NW : 0;
      if ((int)blockIdx.x < n_left) __syncthreads();  // (uniform per workgroup) the last fold has read the exchange areas
    } else {
      const int n_units = n_left * nrb, U = (n_units + gridDim.x * CE_NW - 1) / (gridDim.x * CE_NW);
      first = wid * U;
      stride = 1;
      n_my = min(max(n_units - first, 0), U);
    }
    constexpr int TSX = 48;  // 16 * odd
    static_assert(C::ITEMS * C::ES >= 32 * C::ES + 16 * TSX, "the wave's table-row area holds the leftover tile's working set");
    float* Ex = Es;                 // [16][ES] the leftover tile's table rows
    float* Hx = Es + 16 * C::ES;    // [16][ES] this wave's batch rows
    float* Xx = Hx + 16 * C::ES;    // [16][TSX] transpose scratch
    f4 dEx[C::DT];
#pragma unroll
    for (int dt = 0; dt < C::DT; ++dt) dEx[dt] = f4{0.f, 0.f, 0.f, 0.f};
    // a unit is too short to cover the next one's round trip: rows, lse, coef and target of FOUR units are requested together
    constexpr int HX = 16 * (CH / 4) / 64;  // float4s of a [16, CH] block per lane
    constexpr int GU = 4;
    f4 hx_g[GU][HX];
    float lse_g[GU], cf_g[GU];
    int tgt_g[GU];
    auto prefetch_x = [&](int q, int rb) {
#pragma unroll
      for (int u = 0; u < HX; ++u) {
        const int idx = lane + 64 * u;
        const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
        hx_g[q][u] = f4{0.f, 0.f, 0.f, 0.f};
        if (16 * rb + r < B) hx_g[q][u] = *(const f4*)(P.out + (size_t)(16 * rb + r) * CH + 4 * c4);
      }
      const int row = 16 * rb + c;
      const bool ok = row < B;
      lse_g[q] = (ok && !DIR) ? lse[row] : 0.f;
      cf_g[q] = (ok && !DIR) ? coef[P.coef_is_scalar ? 0 : row] * (P.coef_scale != 0.f ? P.coef_scale : 1.0f) : 0.f;
      tgt_g[q] = ok ? (int)P.target[row] : -1;
    };
    int cur_tile = -1;
    for (int k0 = 0; k0 < n_my; k0 += GU) {
#pragma unroll
      for (int q = 0; q < GU; ++q)
        if (k0 + q < n_my) prefetch_x(q, (first + (k0 + q) * stride) % nrb);
#pragma unroll
      for (int q = 0; q < GU; ++q) {
      if (k0 + q >= n_my) break;
      const int u = first + (k0 + q) * stride, tile = u / nrb, rb = u - tile * nrb;
      const int itx = gridDim.x * CE_NW * C::ITEMS + 16 * tile;
      const size_t vslab = (size_t)(gridDim.x + tile) * B;  // row offset of this tile's slab / partials
      if (tile != cur_tile) {  // (uniform per wave)
        cur_tile = tile;
        for (int idx = lane; idx < 16 * (CH / 4); idx += 64) {
          const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (itx + r < N) v = *(const f4*)(P.table + (size_t)(itx + r) * CH + 4 * c4);
          *(f4*)(Ex + r * C::ES + 4 * c4) = v;
        }
      }
#pragma unroll
      for (int w = 0; w < HX; ++w) {
        const int idx = lane + 64 * w;
        const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
        *(f4*)(Hx + r * C::ES + 4 * c4) = hx_g[q][w];
      }
      const int row = 16 * rb + c;
      const bool ok = row < B;
      const float l2 = lse_g[q] * kLog2e;
      const float cf = cf_g[q];
      const int tgt = ok ? tgt_g[q] - itx : -1;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      float hf[C::KS];
#pragma unroll
      for (int s4 = 0; s4 < C::KS / 4; ++s4) {
        const f4 v = *(const f4*)(Hx + c * C::ES + C::KS * g + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) hf[4 * s4 + e] = v[e];
      }
      f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
      for (int s4 = 0; s4 < C::KS / 4; ++s4) {
        const f4 e4 = *(const f4*)(Ex + c * C::ES + C::KS * g + 4 * s4);
        a0 = mfma16(e4[0], hf[4 * s4 + 0], a0);
        a1 = mfma16(e4[1], hf[4 * s4 + 1], a1);
        a0 = mfma16(e4[2], hf[4 * s4 + 2], a0);
        a1 = mfma16(e4[3], hf[4 * s4 + 3], a1);
      }
      f4 dlx = a0 + a1;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (itx + 4 * g + r >= N) dlx[r] = ACATTN_NEG_INF;
      float m_w = ACATTN_NEG_INF, s_w = 0.f;
      if (DIR) {
        m_w = quad_max(fmaxf(fmaxf(dlx[0], dlx[1]), fmaxf(dlx[2], dlx[3])));
        const float m2 = m_w > ACATTN_NEG_INF ? m_w * kLog2e : 0.f;
        f4 x = dlx * kLog2e - m2;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
        dlx = x;
        s_w = quad_sum((x[0] + x[1]) + (x[2] + x[3]));
      } else {
        f4 x = dlx * kLog2e - l2;
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
        if (tgt >= 0 && tgt < 16 && ((tgt >> 2) & 3) == g) {
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] -= (r == (tgt & 3)) ? 1.0f : 0.0f;
        }
        dlx = x * cf;
      }
      // d out^T (the tile's 16 items) = E^T . dl^T
      f4 dh[C::DT];
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt) dh[dt] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const float* ep = Ex + (4 * g + kk) * C::ES + c;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) dh[dt] = mfma16(ep[16 * dt], dlx[kk], dh[dt]);
      }
      if (ok) {
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) *(f4*)(d_out_slab + (vslab + row) * CH + 16 * dt + 4 * g) = dh[dt];
        if (DIR && g == 0) part[vslab + row] = float2{m_w, s_w};
      }
      if (WITH_TABLE_GRAD) {
        *(f4*)(Xx + c * TSX + 4 * g) = dlx;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int sr = 0; sr < 4; ++sr) {
          const float at = Xx[(4 * sr + g) * TSX + c];
#pragma unroll
          for (int dt = 0; dt < C::DT; ++dt) dEx[dt] = mfma16(at, Hx[(4 * sr + g) * C::ES + 16 * dt + c], dEx[dt]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      }
    }
    if (WITH_TABLE_GRAD && (int)blockIdx.x < n_left) {  // the four waves' shares of the tile's d_table rows meet in the exchange areas
      const int itx = gridDim.x * CE_NW * C::ITEMS + 16 * blockIdx.x;
#pragma unroll
      for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) Xw[(4 * g + r) * C::ES + 16 * dt + c] = dEx[dt][r];
      __syncthreads();
      for (int idx = threadIdx.x; idx < 16 * CH; idx += blockDim.x) {
        const int i = idx / CH, hcol = idx - i * CH;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < CE_NW; ++w) v += X[w * XS + i * C::ES + hcol];
        if (itx + i < N) d_table[(size_t)(itx + i) * CH + hcol] = v;
      }
    }
  }
}

// d_out[i] = sum over workgroups of slab[wg][i]: 32 outputs per workgroup, 8 threads per output (each sums every
// 8th slab with 8 loads in flight), folded through LDS -- the same shape as wgrad_reduce_kernel.
__global__ void __launch_bounds__(256) ce_bwd_reduce_kernel(const float* __restrict__ slab, const int n_slabs,
                                                             const int64_t n_out, float* __restrict__ d_out) {
  const int sl = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + sl;
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  if (i < n_out) {
    const float* ps = slab + i;
    int p = q;
    for (; p + 56 < n_slabs; p += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += ps[(size_t)(p + 8 * u) * n_out];
    }
    for (; p < n_slabs; p += 8) a[0] += ps[(size_t)p * n_out];
  }
  __shared__ float red[8][32];
  red[q][sl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (q == 0 && i < n_out) {
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) v += red[u][sl];
    d_out[i] = v;
  }
}

// Finishes the forward-with-direction sweep for one batch row per workgroup:
//   m = max_wg m_wg,  w_wg = exp(m_wg - m),  S = sum w_wg s_wg,  O = sum w_wg slab_wg[row]
//   lse = m + log S,  row_loss = lse - out_row . E_target,  dir = O / S - E_target   (= d row_loss / d out_row)
template <int CH>
__global__ void __launch_bounds__(256) ce_dir_reduce_kernel(const acattn_ce_problem P, const float2* __restrict__ part,
                                                             const float* __restrict__ slab, const int n_wg,
                                                             float* __restrict__ lse, float* __restrict__ row_loss,
                                                             float* __restrict__ dir) {
  extern __shared__ float wts[];  // [n_wg] weights, then scratch
  __shared__ float red[256];
  const int row = blockIdx.x, B = P.B;
  float m = ACATTN_NEG_INF;
  for (int k = threadIdx.x; k < n_wg; k += 256) m = fmaxf(m, part[(size_t)k * B + row].x);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  float ssum = 0.f;
  for (int k = threadIdx.x; k < n_wg; k += 256) {
    const float2 p = part[(size_t)k * B + row];
    const float w = p.x > ACATTN_NEG_INF ? __builtin_amdgcn_exp2f((p.x - m) * kLog2e) : 0.f;
    wts[k] = w;
    ssum += w * p.y;
  }
  red[threadIdx.x] = ssum;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  const float S = red[0];
  __syncthreads();
  // O[h]: CH / 4 lanes of float4 per slab row, 256 / (CH / 4) slabs in flight per pass
  constexpr int LPR = CH / 4, GROUPS = 256 / LPR;
  const int c4 = threadIdx.x % LPR, grp = threadIdx.x / LPR;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* base = slab + (size_t)row * CH + 4 * c4;
  for (int k = grp; k < n_wg; k += GROUPS) acc += *(const f4*)(base + (size_t)k * B * CH) * wts[k];
  float* facc = wts + n_wg;  // [GROUPS][CH]
  *(f4*)(facc + grp * CH + 4 * c4) = acc;
  __syncthreads();
  if (threadIdx.x < CH) {
    float o = 0.f;
    for (int k = 0; k < GROUPS; ++k) o += facc[k * CH + threadIdx.x];
    const long long tgt_raw = P.target[row];
    const long long tgt = (tgt_raw >= 0 && tgt_raw < P.N) ? tgt_raw : 0;  // see ce_fwd_reduce_kernel
    const float et = P.table[(size_t)tgt * CH + threadIdx.x];
    dir[(size_t)row * CH + threadIdx.x] = o / S - et;
    red[threadIdx.x] = P.out[(size_t)row * CH + threadIdx.x] * et;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float dot = 0.f;
    for (int k = 0; k < CH; ++k) dot += red[k];
    const float l = m + __builtin_amdgcn_logf(S) * kLn2;
    const long long tgt_raw = P.target[row];
    lse[row] = l;
    row_loss[row] = (tgt_raw >= 0 && tgt_raw < P.N) ? l - dot : __builtin_nanf("");
  }
}

constexpr int64_t kSlabLimit = 512ll << 20;  // bytes of d_out slabs above which the backward uses atomics (96 MB until round 3: configs[3] is 136-162 MB and was paying 34 M float atomics per launch)

static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// (CH = 256: one tile per wave -- the wave's table rows, the batch rows and the d_table accumulators are 64 registers
// each per tile, and the backward's LDS stage is 150 KB at one tile)
template <int CH>
constexpr int max_tiles() { return CH <= 64 ? 7 : (CH <= 128 ? 3 : 1); }

// smallest tile count whose single round of (#CUs) workgroups covers N items
template <int CH>
int pick_tiles(int N) {
  const int per_tile = num_cus() * CE_NW * 16;
  const int need = (N + per_tile - 1) / per_tile;
  if (CH <= 64) return need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 6 ? 6 : 7;
  if (CH > 128) return 1;
  return need <= 1 ? 1 : need <= 2 ? 2 : 3;
}

// forward only (no LDS, <= 256 registers): two workgroups fit a CU, so fewer tiles per wave than one round would need
// put two waves on a SIMD and one wave's soft-max arithmetic runs under the other's MFMAs
template <int CH>
int pick_tiles_fwd(int N) {
  static const int forced = getenv("ACATTN_CE_TILES_FWD") ? atoi(getenv("ACATTN_CE_TILES_FWD")) : 0;  // measurements
  if (forced == 1 || forced == 2 || forced == 4 || forced == 6 || forced == 7) return forced;
  return pick_tiles<CH>(N);
}

// The split scheme (ce_fwd_kernel): whole rounds of (CUs x 4 waves x T tiles) + at most one leftover tile per workgroup.
// Hidden 64: T = 6 and one round, for catalogues of 6.x rounds of tiles (the benchmark's 100,000 items -> 256 workgroups +
// 106 leftover tiles instead of 224 workgroups of seven tiles).  Hidden 128: T = 3 (its maximum) and as many whole rounds
// as fit (100,000 items -> 512 workgroups + 106 leftover tiles instead of 521 workgroups, i.e. a third round for nine of
// them).  ACATTN_CE_SPLIT=0 switches it off (measurement).
template <int CH>
constexpr int split_tiles() { return CH == 64 ? 6 : 3; }

template <int CH>
bool split_plan(int N, int& n_wg, int& n_left) {
  static const bool off = getenv("ACATTN_CE_SPLIT") && atoi(getenv("ACATTN_CE_SPLIT")) == 0;
  if ((CH != 64 && CH != 128) || off) return false;
  const int tiles = (N + 15) / 16, per_round = num_cus() * CE_NW;
  if (CH == 64) {
    if (tiles / per_round != 6) return false;
    n_wg = num_cus();
  } else {
    const int rounds = tiles / (per_round * 3);
    if (rounds < 1) return false;
    n_wg = rounds * num_cus();
  }
  n_left = tiles - n_wg * CE_NW * split_tiles<CH>();
  return n_left > 0 && n_left <= n_wg;
}

// [round 4] hidden 64: the sweeps of acattn_ce_bf16.hip (three-way bf16 split of every operand, six bf16 MFMAs per
// product: fp32 accuracy at 0.375 of the matrix time) wherever six tiles per wave cover the catalogue in one round,
// with or without leftover tiles.  g_ce_products: 0 = exact fp32 MFMA everywhere (acattn_full_sort_ce_products, or
// ACATTN_CE_PRODUCTS=fp32), 1 = the default above, 2 = the split sweeps for every catalogue size (tests).
int g_ce_products = -1;
int ce_products() {
  if (g_ce_products < 0) {
    const char* e = getenv("ACATTN_CE_PRODUCTS");
    g_ce_products = !e ? 1 : !strcmp(e, "fp32") ? 0 : !strcmp(e, "bf16x6_all") ? 2 : 1;
  }
  return g_ce_products;
}

template <int CH>
bool ce6_plan(int N, int& n_wg, int& n_left) {
  if (CH != 64 || ce_products() == 0) return false;
  if (split_plan<CH>(N, n_wg, n_left)) return true;
  // ... and beyond: more than 65,536 items always fill the chip with six-tile waves, in 1.x or more rounds (200,000 items:
  // 521 workgroups, three rounds of ~110 us against two seven-tile fp32 rounds of ~265)
  if (ce_products() == 2 || pick_tiles<CH>(N) >= 6) {
    n_wg = (N + CE_NW * 96 - 1) / (CE_NW * 96);
    n_left = 0;
    return true;
  }
  return false;
}

template <int CH>
int64_t ws_bytes_base(const acattn_ce_problem& p);

inline int64_t align256(int64_t x) { return (x + 255) / 256 * 256; }

template <int CH, int NTILES>
int launch_fwd_t(const acattn_ce_problem& p, void* ws, float* lse, float* row_loss, hipStream_t stream) {
  using C = CeCfg<CH, NTILES>;
  const int n_wg = (p.N + CE_NW * C::ITEMS - 1) / (CE_NW * C::ITEMS);
  hipLaunchKernelGGL((ce_fwd_kernel<CH, NTILES>), dim3(n_wg), dim3(64 * CE_NW), 0, stream, p, (float2*)ws);
  hipLaunchKernelGGL((ce_fwd_reduce_kernel<CH>), dim3((p.B + 3) / 4), dim3(256), 0, stream, p, (const float2*)ws,
                     n_wg * CE_NW, lse, row_loss);
  return (int)hipGetLastError();
}

template <int CH, int NTILES>
int launch_bwd_t(const acattn_ce_problem& p, const float* lse, const float* coef, void* ws, float* d_out, float* d_table,
                 hipStream_t stream) {
  using C = CeCfg<CH, NTILES>;
  const int n_wg = (p.N + CE_NW * C::ITEMS - 1) / (CE_NW * C::ITEMS);
  constexpr int TS = C::ITEMS + 16 + ((C::ITEMS / 16 + 1) % 2 ? 0 : 16);
  constexpr int XS = (16 * TS > 16 * C::ES) ? 16 * TS : 16 * C::ES;
  const size_t lds = (size_t)(CE_NW * C::ITEMS * C::ES + 16 * C::ES + CE_NW * XS) * sizeof(float);
  const int64_t n_out = (int64_t)p.B * CH;
  float* slab = (n_wg * n_out * (int64_t)sizeof(float) <= kSlabLimit) ? (float*)ws : nullptr;
  if (!slab) {
    if (const int e = acattn_launch_zero(d_out, (size_t)n_out, stream)) return e;  // (not hipMemsetAsync: acattn_util.hip)
  }
  if (d_table) {
    auto k = ce_bwd_kernel<CH, NTILES, true>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * CE_NW), lds, stream, p, lse, coef, d_out, slab, d_table, (float2*)nullptr, 0);
  } else {
    auto k = ce_bwd_kernel<CH, NTILES, false>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * CE_NW), lds, stream, p, lse, coef, d_out, slab, d_table, (float2*)nullptr, 0);
  }
  if (slab)
    hipLaunchKernelGGL(ce_bwd_reduce_kernel, dim3((unsigned)((n_out + 31) / 32)), dim3(256), 0, stream, slab, n_wg, n_out,
                       d_out);
  return (int)hipGetLastError();
}

template <int CH, int NTILES>
int launch_fwd_dir_t(const acattn_ce_problem& p, void* ws, float* lse, float* row_loss, float* dir, hipStream_t stream) {
  using C = CeCfg<CH, NTILES>;
  const int n_wg = (p.N + CE_NW * C::ITEMS - 1) / (CE_NW * C::ITEMS);
  constexpr int TS = C::ITEMS + 16 + ((C::ITEMS / 16 + 1) % 2 ? 0 : 16);
  constexpr int XS = (16 * TS > 16 * C::ES) ? 16 * TS : 16 * C::ES;
  const size_t lds = (size_t)(CE_NW * C::ITEMS * C::ES + 16 * C::ES + CE_NW * XS) * sizeof(float);
  const int64_t n_out = (int64_t)p.B * CH;
  if (n_wg * n_out * (int64_t)sizeof(float) > kSlabLimit) return -100;
  float* slab = (float*)ws;
  float2* part = (float2*)(slab + (size_t)n_wg * n_out);
  auto k = ce_bwd_kernel<CH, NTILES, false, true>;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * CE_NW), lds, stream, p, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, slab, (float*)nullptr, part, 0);
  const size_t rlds = (size_t)(n_wg + (256 / (CH / 4)) * CH) * sizeof(float);
  hipLaunchKernelGGL((ce_dir_reduce_kernel<CH>), dim3(p.B), dim3(256), rlds, stream, p, (const float2*)part,
                     (const float*)slab, n_wg, lse, row_loss, dir);
  return (int)hipGetLastError();
}

template <int CH>
int launch_fwd_dir(const acattn_ce_problem& p, void* ws, float* lse, float* row_loss, float* dir, hipStream_t stream) {
  if constexpr (CH == 64) {
    int n_wg, n_left;
    const int64_t n_out = (int64_t)p.B * CH;
    if (ce6_plan<CH>(p.N, n_wg, n_left) && (n_wg + n_left) * n_out * (int64_t)sizeof(float) <= kSlabLimit) {
      const int n_slabs = n_wg + n_left;
      float* slab = (float*)ws;
      float2* part = (float2*)(slab + (size_t)n_slabs * n_out);
      void* rows_ws = (char*)ws + align256(ws_bytes_base<CH>(p));
      if (const int e = acattn_launch_ce6_sweep(p, nullptr, nullptr, slab, nullptr, part, rows_ws, n_wg, n_left, true, stream)) return e;
      const size_t rlds = (size_t)(n_slabs + (256 / (CH / 4)) * CH) * sizeof(float);
      hipLaunchKernelGGL((ce_dir_reduce_kernel<CH>), dim3(p.B), dim3(256), rlds, stream, p, (const float2*)part,
                         (const float*)slab, n_slabs, lse, row_loss, dir);
      return (int)hipGetLastError();
    }
  }
  if constexpr (CH == 64 || CH == 128) {
    int n_wg, n_left;
    const int64_t n_out = (int64_t)p.B * CH;
    if (split_plan<CH>(p.N, n_wg, n_left) && (n_wg + n_left) * n_out * (int64_t)sizeof(float) <= kSlabLimit) {
      using C = CeCfg<CH, split_tiles<CH>()>;
      constexpr int TS = C::ITEMS + 16 + ((C::ITEMS / 16 + 1) % 2 ? 0 : 16);
      constexpr int XS = (16 * TS > 16 * C::ES) ? 16 * TS : 16 * C::ES;
      const size_t lds = (size_t)(CE_NW * C::ITEMS * C::ES + 16 * C::ES + CE_NW * XS) * sizeof(float);
      const int n_slabs = n_wg + n_left;
      float* slab = (float*)ws;
      float2* part = (float2*)(slab + (size_t)n_slabs * n_out);
      auto k = ce_bwd_kernel<CH, split_tiles<CH>(), false, true, true>;
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * CE_NW), lds, stream, p, (const float*)nullptr, (const float*)nullptr,
                         (float*)nullptr, slab, (float*)nullptr, part, n_left);
      const size_t rlds = (size_t)(n_slabs + (256 / (CH / 4)) * CH) * sizeof(float);
      hipLaunchKernelGGL((ce_dir_reduce_kernel<CH>), dim3(p.B), dim3(256), rlds, stream, p, (const float2*)part,
                         (const float*)slab, n_slabs, lse, row_loss, dir);
      return (int)hipGetLastError();
    }
  }
  if constexpr (CH > 128) return launch_fwd_dir_t<CH, 1>(p, ws, lse, row_loss, dir, stream);
  else
  switch (pick_tiles<CH>(p.N)) {
    case 1: return launch_fwd_dir_t<CH, 1>(p, ws, lse, row_loss, dir, stream);
    case 2: return launch_fwd_dir_t<CH, 2>(p, ws, lse, row_loss, dir, stream);
    case 3: return launch_fwd_dir_t<CH, (CH <= 64 ? 4 : 3)>(p, ws, lse, row_loss, dir, stream);
    case 4: return launch_fwd_dir_t<CH, (CH <= 64 ? 4 : 3)>(p, ws, lse, row_loss, dir, stream);
    case 6: return launch_fwd_dir_t<CH, (CH <= 64 ? 6 : 3)>(p, ws, lse, row_loss, dir, stream);
    default: return launch_fwd_dir_t<CH, max_tiles<CH>()>(p, ws, lse, row_loss, dir, stream);
  }
}

template <int CH>
int64_t ws_bytes_base(const acattn_ce_problem& p);

// the regular scratch, then (hidden 64) the batch rows' operand images of the split sweeps
template <int CH>
int64_t ws_bytes(const acattn_ce_problem& p) {
  const int64_t base = ws_bytes_base<CH>(p);
  if (CH != 64 || base < 0) return base;
  return align256(base) + acattn_ce6_rows_bytes(p);
}

template <int CH>
int64_t ws_bytes_base(const acattn_ce_problem& p) {
  int64_t six = 0;  // the six-tile plan without leftovers (any catalogue size when forced): slabs + (max, sum-exp) pairs
  if constexpr (CH == 64) {
    const int64_t n_wg6 = (p.N + CE_NW * 96 - 1) / (CE_NW * 96);
    const int64_t b6 = n_wg6 * p.B * (CH * (int64_t)sizeof(float) + (int64_t)sizeof(float2));
    six = std::max(b6 <= kSlabLimit ? b6 : 0, (n_wg6 * 8 + 256) * p.B * (int64_t)sizeof(float2));  // (forward: eight waves' partials per workgroup + leftover tiles')
  }
  if constexpr (CH == 64 || CH == 128) {
    int n_wg, n_left;
    if (split_plan<CH>(p.N, n_wg, n_left)) {  // (sized for both forms: the slab limit may send a launch to the other one)
      const int64_t fwd = (int64_t)(n_wg * CE_NW + n_left) * p.B * (int64_t)sizeof(float2);
      const int64_t bwd = (int64_t)(n_wg + n_left) * p.B * CH * (int64_t)sizeof(float);
      const int64_t dirb = bwd + (int64_t)(n_wg + n_left) * p.B * (int64_t)sizeof(float2);
      const int64_t regular_wg = (p.N + CE_NW * 16 * max_tiles<CH>() - 1) / (CE_NW * 16 * max_tiles<CH>());
      const int64_t regular = std::max(regular_wg * CE_NW * p.B * (int64_t)sizeof(float2),
                                       regular_wg * p.B * (CH * (int64_t)sizeof(float) + (int64_t)sizeof(float2)));
      return std::max(six, std::max(std::max(fwd, regular), dirb <= kSlabLimit ? dirb : bwd <= kSlabLimit ? bwd : 0));
    }
  }
  // forward partials: one (max, sum-exp) pair per (wave, row)
  int tiles = pick_tiles<CH>(p.N);
  if (CH <= 64 && tiles == 3) tiles = 4;
  if (tiles > max_tiles<CH>()) tiles = max_tiles<CH>();
  const int64_t n_wg = (p.N + CE_NW * 16 * tiles - 1) / (CE_NW * 16 * tiles);
  int ftiles = CH > 128 ? 1 : pick_tiles_fwd<CH>(p.N);
  if (CH <= 64 && ftiles == 3) ftiles = 4;
  if (ftiles > max_tiles<CH>()) ftiles = max_tiles<CH>();
  const int64_t n_wg_fwd = (p.N + CE_NW * 16 * ftiles - 1) / (CE_NW * 16 * ftiles);
  const int64_t fwd = n_wg_fwd * CE_NW * p.B * (int64_t)sizeof(float2);
  // backward: one [B, CH] slab of d_out per workgroup (skipped, in favour of atomics, beyond kSlabLimit)
  const int64_t bwd = n_wg * p.B * CH * (int64_t)sizeof(float);
  // forward-with-direction: the same slabs plus one (max, sum-exp) pair per (workgroup, row)
  const int64_t dirb = bwd <= kSlabLimit ? bwd + n_wg * p.B * (int64_t)sizeof(float2) : 0;
  return std::max(six, std::max(fwd, std::max(bwd <= kSlabLimit ? bwd : 0, dirb)));
}

template <int CH>
int launch_fwd(const acattn_ce_problem& p, void* ws, float* lse, float* row_loss, hipStream_t stream) {
  if constexpr (CH == 64) {
    int n_wg, n_left;
    if (ce6_plan<CH>(p.N, n_wg, n_left)) {
      void* rows_ws = (char*)ws + align256(ws_bytes_base<CH>(p));
      if (const int e = acattn_launch_ce6_fwd_sweep(p, (float2*)ws, rows_ws, n_wg, n_left, stream)) return e;
      hipLaunchKernelGGL((ce_fwd_reduce_kernel<CH>), dim3((p.B + 3) / 4), dim3(256), 0, stream, p, (const float2*)ws,
                         n_wg + n_left, lse, row_loss);
      return (int)hipGetLastError();
    }
  }
  if constexpr (CH == 64 || CH == 128) {
    int n_wg, n_left;
    if (split_plan<CH>(p.N, n_wg, n_left)) {
      hipLaunchKernelGGL((ce_fwd_kernel<CH, split_tiles<CH>(), true>), dim3(n_wg), dim3(64 * CE_NW), 0, stream, p, (float2*)ws, n_left);
      hipLaunchKernelGGL((ce_fwd_reduce_kernel<CH>), dim3((p.B + 3) / 4), dim3(256), 0, stream, p, (const float2*)ws,
                         n_wg * CE_NW + n_left, lse, row_loss);
      return (int)hipGetLastError();
    }
  }
  if constexpr (CH > 128) return launch_fwd_t<CH, 1>(p, ws, lse, row_loss, stream);
  else
  switch (pick_tiles_fwd<CH>(p.N)) {
    case 1: return launch_fwd_t<CH, 1>(p, ws, lse, row_loss, stream);
    case 2: return launch_fwd_t<CH, 2>(p, ws, lse, row_loss, stream);
    case 3: return launch_fwd_t<CH, (CH <= 64 ? 4 : 3)>(p, ws, lse, row_loss, stream);
    case 4: return launch_fwd_t<CH, (CH <= 64 ? 4 : 3)>(p, ws, lse, row_loss, stream);
    case 6: return launch_fwd_t<CH, (CH <= 64 ? 6 : 3)>(p, ws, lse, row_loss, stream);
    default: return launch_fwd_t<CH, max_tiles<CH>()>(p, ws, lse, row_loss, stream);
  }
}

template <int CH>
int launch_bwd(const acattn_ce_problem& p, const float* lse, const float* coef, void* ws, float* d_out, float* d_table,
               hipStream_t stream) {
  if constexpr (CH == 64) {
    int n_wg, n_left;
    const int64_t n_out = (int64_t)p.B * CH;
    if (ce6_plan<CH>(p.N, n_wg, n_left) && (n_wg + n_left) * n_out * (int64_t)sizeof(float) <= kSlabLimit) {
      float* slab = (float*)ws;
      void* rows_ws = (char*)ws + align256(ws_bytes_base<CH>(p));
      if (const int e = acattn_launch_ce6_sweep(p, lse, coef, slab, d_table, nullptr, rows_ws, n_wg, n_left, false, stream)) return e;
      return acattn_launch_ce6_onehot_reduce(p, coef, slab, n_wg + n_left, d_out, d_table, stream);
    }
  }
  if constexpr (CH == 64 || CH == 128) {
    int n_wg, n_left;
    const int64_t n_out = (int64_t)p.B * CH;
    if (split_plan<CH>(p.N, n_wg, n_left) && (n_wg + n_left) * n_out * (int64_t)sizeof(float) <= kSlabLimit) {
      using C = CeCfg<CH, split_tiles<CH>()>;
      constexpr int TS = C::ITEMS + 16 + ((C::ITEMS / 16 + 1) % 2 ? 0 : 16);
      constexpr int XS = (16 * TS > 16 * C::ES) ? 16 * TS : 16 * C::ES;
      const size_t lds = (size_t)(CE_NW * C::ITEMS * C::ES + 16 * C::ES + CE_NW * XS) * sizeof(float);
      float* slab = (float*)ws;
      if (d_table) {
        auto k = ce_bwd_kernel<CH, split_tiles<CH>(), true, false, true>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * CE_NW), lds, stream, p, lse, coef, d_out, slab, d_table, (float2*)nullptr, n_left);
      } else {
        auto k = ce_bwd_kernel<CH, split_tiles<CH>(), false, false, true>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * CE_NW), lds, stream, p, lse, coef, d_out, slab, d_table, (float2*)nullptr, n_left);
      }
      hipLaunchKernelGGL(ce_bwd_reduce_kernel, dim3((unsigned)((n_out + 31) / 32)), dim3(256), 0, stream, slab, n_wg + n_left,
                         n_out, d_out);
      return (int)hipGetLastError();
    }
  }
  if constexpr (CH > 128) return launch_bwd_t<CH, 1>(p, lse, coef, ws, d_out, d_table, stream);
  else
  switch (pick_tiles<CH>(p.N)) {
    case 1: return launch_bwd_t<CH, 1>(p, lse, coef, ws, d_out, d_table, stream);
    case 2: return launch_bwd_t<CH, 2>(p, lse, coef, ws, d_out, d_table, stream);
    case 3: return launch_bwd_t<CH, (CH <= 64 ? 4 : 3)>(p, lse, coef, ws, d_out, d_table, stream);
    case 4: return launch_bwd_t<CH, (CH <= 64 ? 4 : 3)>(p, lse, coef, ws, d_out, d_table, stream);
    case 6: return launch_bwd_t<CH, (CH <= 64 ? 6 : 3)>(p, lse, coef, ws, d_out, d_table, stream);
    default: return launch_bwd_t<CH, max_tiles<CH>()>(p, lse, coef, ws, d_out, d_table, stream);
  }
}

}  // namespace

int acattn_ce_products_choice(int mode) {
  const int old = ce_products();
  if (mode >= 0 && mode <= 2) g_ce_products = mode;
  return old;
}

int64_t acattn_ce_ws_bytes(const acattn_ce_problem& p) {
  switch (p.H) {
    case 64: return ws_bytes<64>(p);
    case 128: return ws_bytes<128>(p);
    case 256: return ws_bytes<256>(p);
  }
  return -1;
}

int acattn_launch_ce_fwd(const acattn_ce_problem& p, void* ws, float* lse, float* row_loss, hipStream_t stream) {
  switch (p.H) {
    case 64: return launch_fwd<64>(p, ws, lse, row_loss, stream);
    case 128: return launch_fwd<128>(p, ws, lse, row_loss, stream);
    case 256: return launch_fwd<256>(p, ws, lse, row_loss, stream);
  }
  return -1;
}

int acattn_launch_ce_fwd_dir(const acattn_ce_problem& p, void* ws, float* lse, float* row_loss, float* dir,
                             hipStream_t stream) {
  switch (p.H) {
    case 64: return launch_fwd_dir<64>(p, ws, lse, row_loss, dir, stream);
    case 128: return launch_fwd_dir<128>(p, ws, lse, row_loss, dir, stream);
    case 256: return launch_fwd_dir<256>(p, ws, lse, row_loss, dir, stream);
  }
  return -1;
}

int acattn_launch_ce_bwd(const acattn_ce_problem& p, const float* lse, const float* coef, void* ws, float* d_out,
                         float* d_table, hipStream_t stream) {
  switch (p.H) {
    case 64: return launch_bwd<64>(p, lse, coef, ws, d_out, d_table, stream);
    case 128: return launch_bwd<128>(p, lse, coef, ws, d_out, d_table, stream);
    case 256: return launch_bwd<256>(p, lse, coef, ws, d_out, d_table, stream);
  }
  return -1;
}

#ifdef ACATTN_CE_STAMPS
extern "C" int acattn_debug_ce_stamps(unsigned long long* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ce_stamps), (size_t)n_words * 8);
}
#endif
