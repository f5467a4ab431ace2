// The position-wise tail of one branch of an AC-TSR encoder layer as ONE launch, forward and backward, for gfx950:
//
//     a   = LayerNorm(dropout(dense(ctx)) + x)                  cal_adjusted_outputs, recbole/model/layers.py:681-683
//     out = LayerNorm(dropout(dense_2(gelu(dense_1(a)))) + a)   FeedForward.forward,  layers.py:790-798 (erf-GELU :776-785)
//
// Three small GEMMs (H x H, I x H, H x I with H = 64, I = 256: 147 KB of weights), two LayerNorms and a GELU: as
// separate launches they are six kernels forward and about fifteen backward, each at the launch floor for the
// 512-row tails of the last layer and each a full HBM round trip of [rows, H] / [rows, I] for the 25,600-row tail
// of the first.  Here one WAVE owns 16 (or 32) rows for the whole chain and everything stays in registers:
//
//   * every product is computed TRANSPOSED, out^T = W . in^T, with the exact-fp32 16x16x4 MFMA: the batch row sits on
//     the lane index (lane & 15), so a GEMM's accumulator registers are directly the B operand of the next GEMM, the
//     LayerNorm statistics of a row are 16 register values + one 4-lane reduction, and bias / dropout / residual /
//     GELU are lane-local;
//   * the contraction index is visited in the order the previous accumulator happens to hold it (k-step (t, r) =
//     feature 16t + 4g + r on lane group g), so the weight fragment of an output tile is one 16-byte load per lane:
//     W[16 nt + c][16 t + 4 g ..+3].  Weights are read from L2 (they are 147 KB and every wave reads all of them);
//     the next k-slab's fragments are requested before the current slab's MFMAs;
//   * dense_1 -> GELU -> dense_2 is one loop over the 16-column slabs of the inner dimension: the [16 rows, I]
//     activation never exists as a whole, each slab is stored (the weight-gradient GEMM needs it) and consumed.
//
// Backward (input gradients, LayerNorm parameter partials): the same chain mirrored; the inner activation is
// REBUILT from `a` (one more H x I product per slab instead of a 26 MB read), the transposed weights are gathered
// as dwords straight from the row-major parameters (no transposed copies).  The weight/bias gradients stay with
// acattn_linear_wgrad_grouped, which reads the d_h1 / d_h2 / d_h3 tiles this kernel writes when asked to.
//
// Measured (MI355X, rocprofv3, 25,600 rows / 512 rows; profiles/r02_tail_kernels.txt): forward 38.3 / 14.0 us,
// backward 51.0 / 14.6 us (512 rows: four waves per row block, 17.8 / 24.5 with one), against 6 + ~15 launches of
// 266 / 107 us in total per forward + two backward walks.  The four-wave split on 25,600 rows: 67 / 65 us.
// What bounds it: the fp32 MFMAs (576 forward, 784 backward per 16 rows, 32 cycles each) share the issue port with
// the VALU work between them (GELU, LayerNorm, address arithmetic: about +40 %), and 1,600 row blocks on 1,024 SIMDs
// leave the critical SIMD with two blocks.  Without the stores 37.5 us, without GELU 34.0, without both 33.7, with the
// weight fragments always hitting L1 31.4: neither HBM nor L2 is the limit.
//
// Dropout decisions: row_keep_scale() of acattn_rowops.h, i.e. exactly those of acattn_ln.hip for the same
// (seed, row, column): the fused and the unfused formulation are interchangeable between forward and backward.
#include <stdlib.h>

#include <algorithm>

#include "acattn_common.h"
#include "acattn_rowops.h"
#include "acattn_wstage.h"

int acattn_tail_bwd_partial_rows(int rows);
int acattn_tail_bwd_partial_rows_h(int rows, int H);

namespace {

int g_tail_nb = getenv("ACATTN_TAIL_BLOCKS") ? atoi(getenv("ACATTN_TAIL_BLOCKS")) : 0;  // rows per wave / 16; 0 = by size (measurement hooks: this and acattn_select_layer_tail_blocks)
// four waves per row block while one wave per block would leave most SIMDs without work
bool split_slabs(int rows) {
  static const int limit = getenv("ACATTN_TAIL_SPLIT_ROWS") ? atoi(getenv("ACATTN_TAIL_SPLIT_ROWS")) : 4096;
  return g_tail_nb == 0 && rows <= limit;
}
#ifndef ACATTN_TAIL64_WAVES
#define ACATTN_TAIL64_WAVES 3
#endif
// hidden 64: the staged (weights through LDS) form from this many rows up, unless ACATTN_TAIL_PER_WAVE is set.  Measured
// (tools/tail_time.py, I = 256): 102,400 rows 119 against 140 us forward, 457 against 501 us forward + backward; at the
// 25,600 rows of the L = 50 benchmark both take 40 us forward (1,600 row blocks on 1,024 SIMDs: the second block of the
// busiest SIMD sets the length either way) and the step is 17 us slower with the staged form.
bool staged64(int rows) {
  static const bool per_wave = getenv("ACATTN_TAIL_PER_WAVE") != nullptr;
  static const int from = getenv("ACATTN_TAIL64_STAGED_ROWS") ? atoi(getenv("ACATTN_TAIL64_STAGED_ROWS")) : 32768;
  return !per_wave && g_tail_nb == 0 && rows >= from;
}
// hidden 128 / 256: one row block per wave; four waves per block below this many rows
bool wide_split(int rows) { return rows <= 8192; }
int rows_per_wave(int rows) { return split_slabs(rows) ? 16 : 16 * (g_tail_nb ? g_tail_nb : (rows >= 16384 ? 2 : 1)); }

constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;

// Phi(x) = (1 + erf(x / sqrt 2)) / 2 and exp(-x^2 / 2), branch-free: libm's erff is ~45 instructions with a divergent
// branch per element, and on this chip VALU work does not hide behind the fp32 MFMAs.  Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7 absolute, i.e. at the rounding level of the fp32 result) with the hardware rcp / exp2:
//     erf(z) = 1 - (a1 t + ... + a5 t^5) exp(-z^2),  t = 1 / (1 + p z),  z >= 0
struct PhiExp {
  float phi, e;  // Phi(x), exp(-x^2 / 2)
};
__device__ __forceinline__ PhiExp phi_exp(float x) {
  const float z = fabsf(x) * kInvSqrt2;
  const float t = fast_rcp(fmaf(0.3275911f, z, 1.0f));
  const float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);  // exp(-x^2/2) = 2^(-x^2 log2(e) / 2)
  float q = fmaf(t, 1.061405429f, -1.453152027f);
  q = fmaf(t, q, 1.421413741f);
  q = fmaf(t, q, -0.284496736f);
  q = fmaf(t, q, 0.254829592f);
  const float half_tail = 0.5f * (q * t) * e;  // (1 - erf(z)) / 2
  return PhiExp{x >= 0.f ? 1.0f - half_tail : half_tail, e};
}
__device__ __forceinline__ float gelu_erf(float x) { return x * phi_exp(x).phi; }  // layers.py:776-785
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const PhiExp pe = phi_exp(x);
  return fmaf(x * kInvSqrt2Pi, pe.e, pe.phi);
}

template <int NB>
struct Rows {
  int row[NB];      // clamped (loads)
  int64_t src[NB];  // row of ctx / x (and of d_ctx / d_x) this tail row reads: == row without a source index
  bool ok[NB];      // row < R (stores)
};

template <int NB>
__device__ __forceinline__ Rows<NB> wave_rows(const acattn_tail_problem& P) {
  Rows<NB> w;
  const int c = threadIdx.x & 15, R = P.rows;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int r = (blockIdx.x * NB + nb) * 16 + c;
    w.ok[nb] = r < R;
    w.row[nb] = r < R ? r : R - 1;
    // gather and tail commute (the tail is position-wise): the positions the model reads are picked here instead of
    // by a gather launch per tensor in front of the tail (and a scatter launch per gradient behind it)
    // (positions are clamped into [0, src_L): an out-of-range one -- item_seq_len == 0 gives -1 -- is a caller error that
    // must not become an access outside the sequence)
    w.src[nb] = P.src_index
                    ? (int64_t)(w.row[nb] / P.src_R) * P.src_L + min(max((int)P.src_index[w.row[nb]], 0), P.src_L - 1)
                    : w.row[nb];
  }
  return w;
}

// y = LayerNorm(z * keep + res) in the accumulator layout (lane (c, g), tile t, register r <-> row c, feature 16t+4g+r);
// gamma / beta as the lane's columns of the parameters (requested by the caller, well ahead)
template <int DT>
__device__ __forceinline__ void ln_forward(const f4 (&z)[DT], const f4 (&res)[DT], const f4 (&keep)[DT], const f4 (&gamma)[DT],
                                           const f4 (&beta)[DT], float eps, f4 (&y)[DT], float& mean, float& rstd) {
  constexpr float inv_h = 1.0f / (16 * DT);
  f4 s[DT];
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    s[t] = z[t] * keep[t] + res[t];
    sum += (s[t][0] + s[t][1]) + (s[t][2] + s[t][3]);
  }
  mean = quad_sum(sum) * inv_h;
  float sq = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    s[t] = s[t] - mean;
    sq += (s[t][0] * s[t][0] + s[t][1] * s[t][1]) + (s[t][2] * s[t][2] + s[t][3] * s[t][3]);
  }
  rstd = __builtin_amdgcn_rsqf(quad_sum(sq) * inv_h + eps);
#pragma unroll
  for (int t = 0; t < DT; ++t) y[t] = (s[t] * rstd) * gamma[t] + beta[t];
}

// the same with gamma / beta read where they are used (hidden 128 / 256: no registers to park them in)
template <int DT>
__device__ __forceinline__ void ln_forward_mem(const f4 (&z)[DT], const f4 (&res)[DT], const f4 (&keep)[DT], const float* gamma,
                                               const float* beta, int g, float eps, f4 (&y)[DT], float& mean, float& rstd) {
  constexpr float inv_h = 1.0f / (16 * DT);
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    y[t] = z[t] * keep[t] + res[t];
    sum += (y[t][0] + y[t][1]) + (y[t][2] + y[t][3]);
  }
  mean = quad_sum(sum) * inv_h;
  float sq = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    y[t] = y[t] - mean;
    sq += (y[t][0] * y[t][0] + y[t][1] * y[t][1]) + (y[t][2] * y[t][2] + y[t][3] * y[t][3]);
  }
  rstd = __builtin_amdgcn_rsqf(quad_sum(sq) * inv_h + eps);
#pragma unroll
  for (int t = 0; t < DT; ++t) y[t] = (y[t] * rstd) * *(const f4*)(gamma + 16 * t + 4 * g) + *(const f4*)(beta + 16 * t + 4 * g);
}

__device__ __forceinline__ float hsum4(const f4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

template <int DT>
__device__ __forceinline__ void load_cols(const float* v, int g, f4 (&out)[DT]) {
#pragma unroll
  for (int t = 0; t < DT; ++t) out[t] = *(const f4*)(v + 16 * t + 4 * g);
}

// -----------------------------------------------------------------------------------------------------------------
// forward
// -----------------------------------------------------------------------------------------------------------------
// NW = waves that share one 16-row block (1 or 4).  With few rows (the 512 read positions of the last layer: 32
// blocks) one wave per block leaves the chip empty and the launch lasts as long as one wave's whole chain; with
// NW = 4 every wave forms `a` for itself (64 MFMAs, cheap) and then takes every fourth slab of the inner dimension;
// the four partial products meet in LDS and wave 0 finishes.
template <int H, int I, int NB, int NW = 1>
__global__ void __launch_bounds__(64 * NW) tail_fwd_kernel(const acattn_tail_problem P, const acattn_tail_saved S) {
  static_assert(NW == 1 || NB == 1, "the slab split works on one row block");
  constexpr int DT = H / 16, IT = I / 16, NS = IT / NW;  // NS slabs per wave: mt = wave, wave + NW, ...
  static_assert(NS >= 2, "two slabs are requested ahead");
  const int c = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
  const int wave = NW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const Rows<NB> W = wave_rows<NB>(P);
  const uint64_t step = P.seed_device ? *P.seed_device : 0ull;

  // weight fragments (and the bias) of one 16-column slab of the inner dimension
  struct Slab {
    f4 w1[DT];  // A[m = 16mt+c][k = 16t+4g+r]
    f4 w2[DT];  // A[n = 16t+c][m = 16mt+4g+r]
    f4 b1;
  };
  auto load_slab = [&](int mt, Slab& s) {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      s.w1[t] = *(const f4*)(P.w1 + (size_t)(16 * mt + c) * H + 16 * t + 4 * g);
      s.w2[t] = *(const f4*)(P.w2 + (size_t)(16 * t + c) * I + 16 * mt + 4 * g);
    }
    s.b1 = *(const f4*)(P.bb1 + 16 * mt + 4 * g);
  };

  // ---- every request of the prologue at once: rows, dense weights, LayerNorm parameters, the first two slabs ---------
  f4 cb[NB][DT], res[NB][DT];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      cb[nb][t] = *(const f4*)(P.ctx + (size_t)W.src[nb] * H + 16 * t + 4 * g);
      res[nb][t] = *(const f4*)(P.x + (size_t)W.src[nb] * H + 16 * t + 4 * g);
    }
  f4 wd[DT][DT], bd[DT], g1[DT], b1[DT];
#pragma unroll
  for (int nt = 0; nt < DT; ++nt)
#pragma unroll
    for (int t = 0; t < DT; ++t) wd[nt][t] = *(const f4*)(P.wd + (size_t)(16 * nt + c) * H + 16 * t + 4 * g);
  load_cols<DT>(P.bd, g, bd);
  load_cols<DT>(P.g1, g, g1);
  load_cols<DT>(P.b1, g, b1);
  Slab sl[3];  // the slab in use, the next one (its first product already runs), the one after that (in flight)
  load_slab(wave, sl[0]);
  load_slab(wave + NW, sl[1]);
  PIN_ORDER();

  // ---- h1 = dense(ctx) + bias;  a = LayerNorm(dropout(h1) + x) ---------------------------------------------------------
  f4 h1[NB][DT], a[NB][DT];
#pragma unroll
  for (int nt = 0; nt < DT; ++nt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) h1[nb][nt] = bd[nt];
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < DT; ++nt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) h1[nb][nt] = mfma16(wd[nt][t][r], cb[nb][t][r], h1[nb][nt]);
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    f4 keep[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) keep[t] = row_keep_scale(P.p1, P.keep1, P.seed1 + step, W.row[nb], 4 * t + g, H);
    float mean, rstd;
    ln_forward<DT>(h1[nb], res[nb], keep, g1, b1, P.eps1, a[nb], mean, rstd);
    if (W.ok[nb] && wave == 0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)W.row[nb] * H + 16 * t + 4 * g;
        *(f4*)(S.h1 + o) = h1[nb][t];
        *(f4*)(S.a + o) = a[nb][t];
      }
      if (g == 0) *(float2*)(S.st1 + 2 * (size_t)W.row[nb]) = float2{mean, rstd};
    }
  }

  // ---- h3 = dense_2(gelu(dense_1(a))), one slab at a time ----------------------------------------------------------------
  // Step j: request slab j + 2; first product of slab j + 1 (MFMA) next to the GELU of slab j (VALU: nothing else of
  // this wave could run under those MFMAs); second product of slab j.
  f4 h3[NB][DT], bb2[DT], g2[DT], b2[DT];
  load_cols<DT>(P.bb2, g, bb2);
  load_cols<DT>(P.g2, g, g2);
  load_cols<DT>(P.b2, g, b2);
  auto first_product = [&](const Slab& s, f4 (&h2)[NB][2]) {
    // two partial accumulators: a dependent 16x16x4 chain issues every 40 cycles, alternating every 32
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) h2[nb][0] = h2[nb][1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) h2[nb][r & 1] = mfma16(s.w1[t][r], a[nb][t][r], h2[nb][r & 1]);
  };
  f4 h2[2][NB][2];
  first_product(sl[0], h2[0]);
#pragma unroll
  for (int nt = 0; nt < DT; ++nt)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) h3[nb][nt] = wave == 0 ? bb2[nt] : f4{0.f, 0.f, 0.f, 0.f};
  static_for<NS>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    const int mt = wave + NW * J;
    if (J + 2 < NS) load_slab(mt + 2 * NW, sl[(J + 2) % 3]);
    PIN_ORDER();
    const Slab& cur = sl[J % 3];
    if (J + 1 < NS) first_product(sl[(J + 1) % 3], h2[(J + 1) & 1]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const f4 pre = (h2[J & 1][nb][0] + h2[J & 1][nb][1]) + cur.b1;
      f4 act;
#pragma unroll
      for (int r = 0; r < 4; ++r) act[r] = gelu_erf(pre[r]);
      if (W.ok[nb]) *(f4*)(S.act + (size_t)W.row[nb] * I + 16 * mt + 4 * g) = act;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt) h3[nb][nt] = mfma16(cur.w2[nt][r], act[r], h3[nb][nt]);
    }
    PIN_ORDER();
  });

  if (NW > 1) {  // fold the waves' partial products into wave 0
    __shared__ f4 red[NW > 1 ? NW - 1 : 1][DT][64];
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) red[wave - 1][t][threadIdx.x & 63] = h3[0][t];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w)
#pragma unroll
      for (int t = 0; t < DT; ++t) h3[0][t] += red[w][t][threadIdx.x & 63];
  }

  // ---- out = LayerNorm(dropout(h3) + a) -------------------------------------------------------------------------
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    f4 keep[DT], y[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) keep[t] = row_keep_scale(P.p2, P.keep2, P.seed2 + step, W.row[nb], 4 * t + g, H);
    float mean, rstd;
    ln_forward<DT>(h3[nb], a[nb], keep, g2, b2, P.eps2, y, mean, rstd);
    if (W.ok[nb]) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)W.row[nb] * H + 16 * t + 4 * g;
        *(f4*)(S.h3 + o) = h3[nb][t];
        *(f4*)(S.out + o) = y[t];
      }
      if (g == 0) *(float2*)(S.st2 + 2 * (size_t)W.row[nb]) = float2{mean, rstd};
    }
  }
}

// -----------------------------------------------------------------------------------------------------------------
// backward
// -----------------------------------------------------------------------------------------------------------------
// dz = rstd (g gamma - mean(g gamma) - xhat mean(g gamma xhat)); partial sums of dgamma = g xhat, dbeta = g over the
// wave's rows go to part[0 .. 2H) (reduced over the 16 row lanes, written by lane c == 0 of every group)
template <int DT>
__device__ __forceinline__ void ln_backward(const f4 (&z)[DT], const f4 (&res)[DT], const f4 (&keep)[DT], const float* gamma,
                                            float mean, float rstd, const f4 (&dy)[DT], bool ok, int g, f4 (&dz)[DT],
                                            f4 (&acc_g)[DT], f4 (&acc_b)[DT]) {
  constexpr float inv_h = 1.0f / (16 * DT);
  f4 xh[DT], gg[DT];
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    xh[t] = ((z[t] * keep[t] + res[t]) - mean) * rstd;
    gg[t] = dy[t] * *(const f4*)(gamma + 16 * t + 4 * g);
    m1 += (gg[t][0] + gg[t][1]) + (gg[t][2] + gg[t][3]);
    m2 += (gg[t][0] * xh[t][0] + gg[t][1] * xh[t][1]) + (gg[t][2] * xh[t][2] + gg[t][3] * xh[t][3]);
    if (ok) {
      acc_g[t] += dy[t] * xh[t];
      acc_b[t] += dy[t];
    }
  }
  m1 = quad_sum(m1) * inv_h;
  m2 = quad_sum(m2) * inv_h;
#pragma unroll
  for (int t = 0; t < DT; ++t) dz[t] = (gg[t] - m1 - xh[t] * m2) * rstd;
}

template <int DT>
__device__ __forceinline__ void store_partial(float* part, const f4 (&acc)[DT], int c, int g) {
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    f4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = dpp_row_sum(acc[t][r]);  // over the 16 rows of the lane group
    if (c == 0) *(f4*)(part + 16 * t + 4 * g) = v;               // every lane of the row holds the sum
  }
}

// a selected position may be picked more than once (AcBERT4Rec pads its masked_index with position 0,
// acbert4rec.py:130-140): with a row selection the (zero-filled) gradient rows are accumulated
__device__ __forceinline__ void store_grad(float* dst, f4 v, bool accumulate) {
  if (accumulate) {
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(dst + e, v[e]);
  } else {
    *(f4*)dst = v;
  }
}

template <int H, int I, int NB, int NW = 1>
__global__ void __launch_bounds__(64 * NW) tail_bwd_kernel(const acattn_tail_problem P, const acattn_tail_saved S,
                                                           const acattn_tail_bwd_io IO) {
  static_assert(NW == 1 || NB == 1, "the slab split works on one row block");
  constexpr int DT = H / 16, IT = I / 16;
  const int c = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
  const int wave = NW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;  // see the forward
  const Rows<NB> W = wave_rows<NB>(P);
  const uint64_t step = P.seed_device ? *P.seed_device : 0ull;
  float* part = IO.dgb_part ? IO.dgb_part + (size_t)blockIdx.x * 4 * H : nullptr;

  // ---- through the second LayerNorm: d h3 (after the dropout), d a (residual share) -------------------------------
  f4 a[NB][DT], dh3[NB][DT], da[NB][DT];
  {
    f4 acc_g[DT], acc_b[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc_g[t] = acc_b[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      f4 z[DT], keep[DT], dy[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)W.row[nb] * H + 16 * t + 4 * g;
        z[t] = *(const f4*)(S.h3 + o);
        a[nb][t] = *(const f4*)(S.a + o);
        dy[t] = *(const f4*)(IO.d_out + o);
        keep[t] = row_keep_scale(P.p2, P.keep2, P.seed2 + step, W.row[nb], 4 * t + g, H);
      }
      const float2 st = *(const float2*)(S.st2 + 2 * (size_t)W.row[nb]);
      ln_backward<DT>(z, a[nb], keep, P.g2, st.x, st.y, dy, W.ok[nb], g, da[nb], acc_g, acc_b);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        dh3[nb][t] = da[nb][t] * keep[t];
        if (IO.d_h3 && W.ok[nb] && wave == 0) *(f4*)(IO.d_h3 + (size_t)W.row[nb] * H + 16 * t + 4 * g) = dh3[nb][t];
        if (wave > 0) da[nb][t] = f4{0.f, 0.f, 0.f, 0.f};  // the residual share of d a travels with wave 0
      }
    }
    if (part && wave == 0) {
      store_partial<DT>(part + 2 * H, acc_g, c, g);
      store_partial<DT>(part + 3 * H, acc_b, c, g);
    }
  }

  // ---- d a += W1^T (gelu'(h2) * (W2^T d h3)), slab by slab; h2 rebuilt from a --------------------------------------
  struct Slab {
    f4 w1[DT];    // dense_1 rows of the slab (rebuilds h2):        A[m = 16mt+c][k = 16t+4g+r]
    f4 w2t[DT];   // dense_2^T: A[m = 16mt+c][n = 16t+4g+r] = W2[n][m]   (dword gathers)
    f4 w1t[DT];   // dense_1^T: A[k = 16nt+c][m = 16mt+4g+r] = W1[m][k]  (dword gathers)
    f4 b1;
  };
  auto load_slab = [&](int mt, Slab& s) {
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      s.w1[t] = *(const f4*)(P.w1 + (size_t)(16 * mt + c) * H + 16 * t + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s.w2t[t][r] = P.w2[(size_t)(16 * t + 4 * g + r) * I + 16 * mt + c];
        s.w1t[t][r] = P.w1[(size_t)(16 * mt + 4 * g + r) * H + 16 * t + c];
      }
    }
    s.b1 = *(const f4*)(P.bb1 + 16 * mt + 4 * g);
  };
  Slab cur, nxt;
  load_slab(wave, cur);
#pragma unroll 2
  for (int mt = wave; mt < IT; mt += NW) {
    load_slab(mt + NW < IT ? mt + NW : mt, nxt);  // one slab ahead (see the forward)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      f4 h2[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, dact[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h2[r & 1] = mfma16(cur.w1[t][r], a[nb][t][r], h2[r & 1]);
          dact[r & 1] = mfma16(cur.w2t[t][r], dh3[nb][t][r], dact[r & 1]);
        }
      const f4 pre = (h2[0] + h2[1]) + cur.b1, dac = dact[0] + dact[1];
      f4 dh2;
#pragma unroll
      for (int r = 0; r < 4; ++r) dh2[r] = dac[r] * gelu_erf_grad(pre[r]);
      if (IO.d_h2 && W.ok[nb]) *(f4*)(IO.d_h2 + (size_t)W.row[nb] * I + 16 * mt + 4 * g) = dh2;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt) da[nb][nt] = mfma16(cur.w1t[nt][r], dh2[r], da[nb][nt]);
    }
    cur = nxt;
  }

  if (NW > 1) {  // fold the waves' shares of d a into wave 0
    __shared__ f4 red[NW > 1 ? NW - 1 : 1][DT][64];
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) red[wave - 1][t][threadIdx.x & 63] = da[0][t];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w)
#pragma unroll
      for (int t = 0; t < DT; ++t) da[0][t] += red[w][t][threadIdx.x & 63];
  }

  // ---- through the first LayerNorm: d h1, d x; then d ctx = d h1 . Wd ---------------------------------------------
  f4 dh1[NB][DT];
  {
    f4 acc_g[DT], acc_b[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) acc_g[t] = acc_b[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      f4 z[DT], res[DT], keep[DT], dz[DT];
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)W.row[nb] * H + 16 * t + 4 * g;
        z[t] = *(const f4*)(S.h1 + o);
        res[t] = *(const f4*)(P.x + (size_t)W.src[nb] * H + 16 * t + 4 * g);
        keep[t] = row_keep_scale(P.p1, P.keep1, P.seed1 + step, W.row[nb], 4 * t + g, H);
      }
      const float2 st = *(const float2*)(S.st1 + 2 * (size_t)W.row[nb]);
      ln_backward<DT>(z, res, keep, P.g1, st.x, st.y, da[nb], W.ok[nb], g, dz, acc_g, acc_b);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)W.row[nb] * H + 16 * t + 4 * g;
        dh1[nb][t] = dz[t] * keep[t];
        if (W.ok[nb]) {
          if (IO.d_x) store_grad(IO.d_x + (size_t)W.src[nb] * H + 16 * t + 4 * g, dz[t], P.src_index != nullptr);
          if (IO.d_h1) *(f4*)(IO.d_h1 + o) = dh1[nb][t];
        }
      }
    }
    if (part) {
      store_partial<DT>(part, acc_g, c, g);
      store_partial<DT>(part + H, acc_b, c, g);
    }
  }
  if (IO.d_ctx) {
    f4 dc[NB][DT];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) dc[nb][nt] = f4{0.f, 0.f, 0.f, 0.f};
    f4 wdt[DT][DT];  // dense^T: A[k = 16nt+c][n = 16t+4g+r] = Wd[n][k]
#pragma unroll
    for (int nt = 0; nt < DT; ++nt)
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) wdt[nt][t][r] = P.wd[(size_t)(16 * t + 4 * g + r) * H + 16 * nt + c];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) dc[nb][nt] = mfma16(wdt[nt][t][r], dh1[nb][t][r], dc[nb][nt]);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      if (W.ok[nb])
#pragma unroll
        for (int nt = 0; nt < DT; ++nt) store_grad(IO.d_ctx + (size_t)W.src[nb] * H + 16 * nt + 4 * g, dc[nb][nt], P.src_index != nullptr);
  }
}


// =====================================================================================================================
// Hidden 128 / 256 (BASELINE configs[3], [4]) [round 3].  The kernels above keep the dense weights ([H, H] as MFMA
// fragments: (H/16)^2 float4 per lane) and three slabs of the feed-forward weights in registers; at H = 128 that is 256 +
// 204 registers.  Here every weight STREAMS, one 16-row tile / one inner slab at a time, each requested one MFMA group
// (32-64 instructions) before its use and two waves per SIMD covering the rest of the latency; one wave owns 16 rows
// (NB = 1), the LayerNorm parameters are read where they are used.  The backward reads TRANSPOSED copies of the three
// weight matrices from a caller workspace (acattn_tail_bwd_io.workspace, filled by tail_transpose_kernel in front of
// it): 16-byte fragment loads instead of 64 dword gathers per slab.
// ---------------------------------------------------------------------------------------------------------------------
// out tile nt of  W[n_out = 16 DT][16 DT] . in^T  (+ bias), tiles emitted in order; weights one tile ahead
template <int DT, class Emit>
__device__ __forceinline__ void stream_square(const float* w, const float* bias, int c, int g, const f4 (&in)[DT], Emit&& emit) {
  constexpr int H = 16 * DT;
  f4 wbuf[2][DT], bb[2];
  auto load = [&](int nt, f4 (&dst)[DT], f4& b) {
    b = bias ? *(const f4*)(bias + 16 * nt + 4 * g) : f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DT; ++t) dst[t] = *(const f4*)(w + (size_t)(16 * nt + c) * H + 16 * t + 4 * g);
  };
  load(0, wbuf[0], bb[0]);
  static_for<DT>([&](auto k) {
    constexpr int NT = decltype(k)::value;
    if constexpr (NT + 1 < DT) load(NT + 1, wbuf[(NT + 1) & 1], bb[(NT + 1) & 1]);
    PIN_ORDER();
    f4 acc[2] = {bb[NT & 1], f4{0.f, 0.f, 0.f, 0.f}};  // two partial accumulators (see first_product above)
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r & 1] = mfma16(wbuf[NT & 1][t][r], in[t][r], acc[r & 1]);
    emit(k, acc[0] + acc[1]);
  });
}

template <int H, int I, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 1 ? 2 : 1) tail_wide_fwd_kernel(const acattn_tail_problem P, const acattn_tail_saved S) {
  constexpr int DT = H / 16, IT = I / 16, NS = IT / NW;
  static_assert(NS >= 2, "two slabs are requested ahead");
  const int c = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
  const int wave = NW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const Rows<1> W = wave_rows<1>(P);
  const uint64_t step = P.seed_device ? *P.seed_device : 0ull;
  const int row = W.row[0];
  const bool ok = W.ok[0];

  // ---- h1 = dense(ctx) + bias;  a = LayerNorm(dropout(h1) + x) ---------------------------------------------------------
  f4 a[DT];
  {
    f4 cb[DT], res[DT], h1[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      cb[t] = *(const f4*)(P.ctx + (size_t)W.src[0] * H + 16 * t + 4 * g);
      res[t] = *(const f4*)(P.x + (size_t)W.src[0] * H + 16 * t + 4 * g);
    }
    stream_square<DT>(P.wd, P.bd, c, g, cb, [&](auto k, f4 v) { h1[decltype(k)::value] = v; });
    f4 keep[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) keep[t] = row_keep_scale(P.p1, P.keep1, P.seed1 + step, row, 4 * t + g, H);
    float mean, rstd;
    ln_forward_mem<DT>(h1, res, keep, P.g1, P.b1, g, P.eps1, a, mean, rstd);
    if (ok && wave == 0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)row * H + 16 * t + 4 * g;
        *(f4*)(S.h1 + o) = h1[t];
        *(f4*)(S.a + o) = a[t];
      }
      if (g == 0) *(float2*)(S.st1 + 2 * (size_t)row) = float2{mean, rstd};
    }
  }

  // ---- h3 = dense_2(gelu(dense_1(a))), one inner slab (16 columns) at a time ------------------------------------------------
  // slab j of this wave = inner tile mt = wave + NW j.  Step j: request w1 of slab j + 2 (its buffer was released by the
  // first product of slab j, done in step j - 1); first product of slab j + 1 next to the GELU of slab j; second product
  // of slab j; request w2 of slab j + 1 into the ONE w2 buffer (a second one does not fit under 256 registers: its lead
  // is the next step's first product, the SIMD's other wave covers the rest).
  f4 w1b[2][DT], w2b[DT], b1b[3];
  auto load_w1 = [&](int mt, f4 (&dst)[DT], f4& b) {
    b = *(const f4*)(P.bb1 + 16 * mt + 4 * g);
#pragma unroll
    for (int t = 0; t < DT; ++t) dst[t] = *(const f4*)(P.w1 + (size_t)(16 * mt + c) * H + 16 * t + 4 * g);
  };
  auto load_w2 = [&](int mt, f4 (&dst)[DT]) {
#pragma unroll
    for (int t = 0; t < DT; ++t) dst[t] = *(const f4*)(P.w2 + (size_t)(16 * t + c) * I + 16 * mt + 4 * g);
  };
  load_w1(wave, w1b[0], b1b[0]);
  load_w1(wave + NW, w1b[1], b1b[1]);
  load_w2(wave, w2b);
  f4 h3[DT];
#pragma unroll
  for (int nt = 0; nt < DT; ++nt) h3[nt] = wave == 0 ? *(const f4*)(P.bb2 + 16 * nt + 4 * g) : f4{0.f, 0.f, 0.f, 0.f};
  PIN_ORDER();
  auto first_product = [&](const f4 (&w1)[DT], f4 (&h2)[2]) {
    h2[0] = h2[1] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) h2[r & 1] = mfma16(w1[t][r], a[t][r], h2[r & 1]);
  };
  f4 h2[2][2];
  first_product(w1b[0], h2[0]);
  static_for<NS>([&](auto jc) {
    constexpr int J = decltype(jc)::value;
    const int mt = wave + NW * J;
    if constexpr (J + 2 < NS) load_w1(mt + 2 * NW, w1b[J & 1], b1b[(J + 2) % 3]);
    PIN_ORDER();
    if constexpr (J + 1 < NS) first_product(w1b[(J + 1) & 1], h2[(J + 1) & 1]);
    const f4 pre = (h2[J & 1][0] + h2[J & 1][1]) + b1b[J % 3];
    f4 act, dact;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const PhiExp pe = phi_exp(pre[r]);
      act[r] = pre[r] * pe.phi;
      dact[r] = fmaf(pre[r] * kInvSqrt2Pi, pe.e, pe.phi);
    }
    if (ok) {
      *(f4*)(S.act + (size_t)row * I + 16 * mt + 4 * g) = act;
      // gelu'(h2): at this width rebuilding h2 in the backward is a fourth of its matrix work, 2 x 4 I bytes per row are not
      if (S.gelu_grad) *(f4*)(S.gelu_grad + (size_t)row * I + 16 * mt + 4 * g) = dact;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) h3[nt] = mfma16(w2b[nt][r], act[r], h3[nt]);
    PIN_ORDER();
    if constexpr (J + 1 < NS) load_w2(mt + NW, w2b);
  });

  if (NW > 1) {  // fold the waves' partial products into wave 0
    __shared__ f4 red[NW > 1 ? NW - 1 : 1][DT][64];
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) red[wave - 1][t][threadIdx.x & 63] = h3[t];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w)
#pragma unroll
      for (int t = 0; t < DT; ++t) h3[t] += red[w][t][threadIdx.x & 63];
  }

  // ---- out = LayerNorm(dropout(h3) + a) -------------------------------------------------------------------------
  {
    f4 keep[DT], y[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) keep[t] = row_keep_scale(P.p2, P.keep2, P.seed2 + step, row, 4 * t + g, H);
    float mean, rstd;
    ln_forward_mem<DT>(h3, a, keep, P.g2, P.b2, g, P.eps2, y, mean, rstd);
    if (ok) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)row * H + 16 * t + 4 * g;
        *(f4*)(S.h3 + o) = h3[t];
        *(f4*)(S.out + o) = y[t];
      }
      if (g == 0) *(float2*)(S.st2 + 2 * (size_t)row) = float2{mean, rstd};
    }
  }
}

// transposed copies for the wide backward: ws = [ W1^T [H][I] | W2^T [I][H] | Wd^T [H][H] ]
__global__ void __launch_bounds__(256) tail_transpose_kernel(const float* w1, const float* w2, const float* wd, int H, int I, float* ws) {
  __shared__ float tile[16][17];
  const int mtx = blockIdx.z;
  const float* src = mtx == 0 ? w1 : mtx == 1 ? w2 : wd;
  const int R = mtx == 0 ? I : H, Cn = mtx == 1 ? I : H;  // src is [R][Cn], dst [Cn][R]
  float* dst = ws + (mtx == 0 ? 0 : mtx == 1 ? (size_t)H * I : 2 * (size_t)H * I);
  const int r0 = blockIdx.x * 16, c0 = blockIdx.y * 16;
  if (r0 >= R || c0 >= Cn) return;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  tile[ty][tx] = src[(size_t)(r0 + ty) * Cn + c0 + tx];
  __syncthreads();
  dst[(size_t)(c0 + ty) * R + r0 + tx] = tile[tx][ty];
}

// LayerNorm backward of one row block without the two accumulator arrays of ln_backward (one row block per wave: the
// partial sums of dgamma / dbeta are the block's own): the normalised value is formed twice instead of being kept
template <int DT>
__device__ __forceinline__ void ln_backward_wide(const f4 (&z)[DT], const f4 (&res)[DT], const f4 (&keep)[DT], const float* gamma,
                                                 float mean, float rstd, const f4 (&dy)[DT], bool ok, bool write_part, int c, int g,
                                                 f4 (&dz)[DT], float* part_g, float* part_b) {
  constexpr float inv_h = 1.0f / (16 * DT);
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const f4 xh = ((z[t] * keep[t] + res[t]) - mean) * rstd;
    const f4 gg = dy[t] * *(const f4*)(gamma + 16 * t + 4 * g);
    m1 += (gg[0] + gg[1]) + (gg[2] + gg[3]);
    m2 += (gg[0] * xh[0] + gg[1] * xh[1]) + (gg[2] * xh[2] + gg[3] * xh[3]);
    if (part_g) {
      f4 pg, pb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pg[r] = dpp_row_sum(ok ? dy[t][r] * xh[r] : 0.f);  // over the 16 rows of the lane group
        pb[r] = dpp_row_sum(ok ? dy[t][r] : 0.f);
      }
      if (c == 0 && write_part) {
        *(f4*)(part_g + 16 * t + 4 * g) = pg;
        *(f4*)(part_b + 16 * t + 4 * g) = pb;
      }
    }
  }
  m1 = quad_sum(m1) * inv_h;
  m2 = quad_sum(m2) * inv_h;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const f4 xh = ((z[t] * keep[t] + res[t]) - mean) * rstd;
    const f4 gg = dy[t] * *(const f4*)(gamma + 16 * t + 4 * g);
    dz[t] = (gg - m1 - xh * m2) * rstd;
  }
}

// SAVED: gelu'(h2) comes from acattn_tail_saved.gelu_grad instead of a rebuilt h2
template <int H, int I, int NW, bool SAVED>
__global__ void __launch_bounds__(64 * NW, NW == 1 ? 2 : 1) tail_wide_bwd_kernel(const acattn_tail_problem P, const acattn_tail_saved S,
                                                                                 const acattn_tail_bwd_io IO, const float* __restrict__ ws) {
  constexpr int DT = H / 16, IT = I / 16;
  const int c = threadIdx.x & 15, g = (threadIdx.x >> 4) & 3;
  const int wave = NW > 1 ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0;
  const Rows<1> W = wave_rows<1>(P);
  const uint64_t step = P.seed_device ? *P.seed_device : 0ull;
  const int row = W.row[0];
  const bool ok = W.ok[0];
  float* part = IO.dgb_part ? IO.dgb_part + (size_t)blockIdx.x * 4 * H : nullptr;
  const float *w1T = ws, *w2T = ws + (size_t)H * I, *wdT = ws + 2 * (size_t)H * I;

  // ---- through the second LayerNorm: d h3 (after the dropout), d a (residual share) -------------------------------
  f4 a[DT], dh3[DT], da[DT];
  {
    f4 z[DT], keep[DT], dy[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const size_t o = (size_t)row * H + 16 * t + 4 * g;
      z[t] = *(const f4*)(S.h3 + o);
      a[t] = *(const f4*)(S.a + o);
      dy[t] = *(const f4*)(IO.d_out + o);
      keep[t] = row_keep_scale(P.p2, P.keep2, P.seed2 + step, row, 4 * t + g, H);
    }
    const float2 st = *(const float2*)(S.st2 + 2 * (size_t)row);
    ln_backward_wide<DT>(z, a, keep, P.g2, st.x, st.y, dy, ok, wave == 0, c, g, da, part ? part + 2 * H : nullptr,
                         part ? part + 3 * H : nullptr);
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      dh3[t] = da[t] * keep[t];
      if (IO.d_h3 && ok && wave == 0) *(f4*)(IO.d_h3 + (size_t)row * H + 16 * t + 4 * g) = dh3[t];
      if (wave > 0) da[t] = f4{0.f, 0.f, 0.f, 0.f};  // the residual share of d a travels with wave 0
    }
  }

  // ---- d a += W1^T (gelu'(h2) * (W2^T d h3)), slab by slab; h2 rebuilt from a --------------------------------------
  // per slab: [request W1^T fragments] [h2 and d act: 2 x 4 DT MFMAs] [request the next slab's W1 / W2^T] [GELU'] [d a: 4 DT]
  {
    f4 w1[SAVED ? 1 : DT], w2t[DT], w1t[DT], b1;  // (b1: the bias of the slab, or its saved gelu')
    auto load_pair = [&](int mt) {
      b1 = SAVED ? *(const f4*)(S.gelu_grad + (size_t)row * I + 16 * mt + 4 * g) : *(const f4*)(P.bb1 + 16 * mt + 4 * g);
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        if constexpr (!SAVED) w1[t] = *(const f4*)(P.w1 + (size_t)(16 * mt + c) * H + 16 * t + 4 * g);  // A[m = 16mt+c][k = 16t+4g+r]
        w2t[t] = *(const f4*)(w2T + (size_t)(16 * mt + c) * H + 16 * t + 4 * g);   // A[m = 16mt+c][n = 16t+4g+r] = W2[n][m]
      }
    };
    load_pair(wave);
    for (int mt = wave; mt < IT; mt += NW) {
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) w1t[nt] = *(const f4*)(w1T + (size_t)(16 * nt + c) * I + 16 * mt + 4 * g);  // A[k][m] = W1[m][k]
      PIN_ORDER();
      f4 h2[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, dact[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (!SAVED) h2[r & 1] = mfma16(w1[t][r], a[t][r], h2[r & 1]);
          dact[r & 1] = mfma16(w2t[t][r], dh3[t][r], dact[r & 1]);
        }
      const f4 pre = (h2[0] + h2[1]) + b1, dac = dact[0] + dact[1];
      PIN_ORDER();
      load_pair(mt + NW < IT ? mt + NW : mt);
      PIN_ORDER();
      f4 dh2;
#pragma unroll
      for (int r = 0; r < 4; ++r) dh2[r] = dac[r] * (SAVED ? pre[r] : gelu_erf_grad(pre[r]));
      if (IO.d_h2 && ok) *(f4*)(IO.d_h2 + (size_t)row * I + 16 * mt + 4 * g) = dh2;
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt) da[nt] = mfma16(w1t[nt][r], dh2[r], da[nt]);
    }
  }

  if (NW > 1) {  // fold the waves' shares of d a into wave 0
    __shared__ f4 red[NW > 1 ? NW - 1 : 1][DT][64];
    if (wave > 0) {
#pragma unroll
      for (int t = 0; t < DT; ++t) red[wave - 1][t][threadIdx.x & 63] = da[t];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w)
#pragma unroll
      for (int t = 0; t < DT; ++t) da[t] += red[w][t][threadIdx.x & 63];
  }

  // ---- through the first LayerNorm: d h1, d x; then d ctx = d h1 . Wd ---------------------------------------------
  f4 dh1[DT];
  {
    f4 z[DT], res[DT], keep[DT], dz[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const size_t o = (size_t)row * H + 16 * t + 4 * g;
      z[t] = *(const f4*)(S.h1 + o);
      res[t] = *(const f4*)(P.x + (size_t)W.src[0] * H + 16 * t + 4 * g);
      keep[t] = row_keep_scale(P.p1, P.keep1, P.seed1 + step, row, 4 * t + g, H);
    }
    const float2 st = *(const float2*)(S.st1 + 2 * (size_t)row);
    ln_backward_wide<DT>(z, res, keep, P.g1, st.x, st.y, da, ok, true, c, g, dz, part, part ? part + H : nullptr);
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const size_t o = (size_t)row * H + 16 * t + 4 * g;
      dh1[t] = dz[t] * keep[t];
      if (ok) {
        if (IO.d_x) store_grad(IO.d_x + (size_t)W.src[0] * H + 16 * t + 4 * g, dz[t], P.src_index != nullptr);
        if (IO.d_h1) *(f4*)(IO.d_h1 + o) = dh1[t];
      }
    }
  }
  if (IO.d_ctx) {
    // dense^T: A[k = 16nt+c][n = 16t+4g+r] = Wd[n][k] = WdT[16nt+c][16t+4g+r]
    stream_square<DT>(wdT, nullptr, c, g, dh1, [&](auto k, f4 v) {
      if (ok) store_grad(IO.d_ctx + (size_t)W.src[0] * H + 16 * decltype(k)::value + 4 * g, v, P.src_index != nullptr);
    });
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Many rows (> 8192): the same chains with every weight chunk staged through LDS once per workgroup of four waves
// (acattn_wstage.h; measurement above proj_staged_fwd_kernel in acattn_proj.hip: a CU's vector memory path, not the
// matrix pipe, bounded the per-wave stream).  A chunk is what one MFMA group of 32 reads: a 16-row tile of Wd, a slab of
// dense_1 (with its bias), the same slab's columns of dense_2; they are consumed in exactly the order they are asked for.
// ---------------------------------------------------------------------------------------------------------------------
template <int NB>
__device__ __forceinline__ Rows<NB> wg_tail_rows(const acattn_tail_problem& P, int wave) {
  Rows<NB> w;
  const int c = threadIdx.x & 15, R = P.rows;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int r = ((blockIdx.x * NWV + wave) * NB + nb) * 16 + c;
    w.ok[nb] = r < R;
    w.row[nb] = r < R ? r : R - 1;
    w.src[nb] = P.src_index
                    ? (int64_t)(w.row[nb] / P.src_R) * P.src_L + min(max((int)P.src_index[w.row[nb]], 0), P.src_L - 1)
                    : w.row[nb];
  }
  return w;
}

template <class Stage>
__device__ __forceinline__ void stage_init(Stage& st, f4* lds) {
  st.lds = lds;
  st.par = 1;  // the first commit fills buffer 0 and flips back to it
  st.lane = threadIdx.x & 63;
  st.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  st.c = threadIdx.x & 15;
  st.g = (threadIdx.x >> 4) & 3;
  st.has_bias = false;
}

template <int H, int I>
__global__ void __launch_bounds__(64 * NWV, 2) tail_staged_fwd_kernel(const acattn_tail_problem P, const acattn_tail_saved S) {
  constexpr int DT = H / 16, IT = I / 16;
  static_assert(DT == KT, "hidden 128");
  __shared__ f4 stage_lds[2 * STAGE_F4];
  WeightStage st;
  stage_init(st, stage_lds);
  const int c = st.c, g = st.g;
  const Rows<1> W = wg_tail_rows<1>(P, st.wave);  // (a wave past the last row keeps walking: the barriers need it)
  const uint64_t step = P.seed_device ? *P.seed_device : 0ull;
  const int row = W.row[0];
  const bool ok = W.ok[0];
  st.request(P.wd, H, P.bd, H, 0, 0);

  // ---- h1 = dense(ctx) + bias;  a = LayerNorm(dropout(h1) + x) ---------------------------------------------------------
  f4 a[DT];
  {
    f4 cb[DT], res[DT], h1[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      cb[t] = *(const f4*)(P.ctx + (size_t)W.src[0] * H + 16 * t + 4 * g);
      res[t] = *(const f4*)(P.x + (size_t)W.src[0] * H + 16 * t + 4 * g);
    }
    st.commit();
    static_for<DT>([&](auto k) {
      constexpr int NT = decltype(k)::value;
      f4 frag[KT], b;
      st.fetch(frag, b);
      if constexpr (NT + 1 < DT) st.request(P.wd, H, P.bd, H, NT + 1, 0); else st.request(P.w1, H, P.bb1, I, 0, 0);
      PIN_ORDER();
      f4 acc[2] = {b, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r & 1] = mfma16(frag[t][r], cb[t][r], acc[r & 1]);
      h1[NT] = acc[0] + acc[1];
      PIN_ORDER();
      st.commit();
    });
    f4 keep[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) keep[t] = row_keep_scale(P.p1, P.keep1, P.seed1 + step, row, 4 * t + g, H);
    float mean, rstd;
    ln_forward_mem<DT>(h1, res, keep, P.g1, P.b1, g, P.eps1, a, mean, rstd);
    if (ok) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)row * H + 16 * t + 4 * g;
        *(f4*)(S.h1 + o) = h1[t];
        *(f4*)(S.a + o) = a[t];
      }
      if (g == 0) *(float2*)(S.st1 + 2 * (size_t)row) = float2{mean, rstd};
    }
  }

  // ---- h3 = dense_2(gelu(dense_1(a))), one inner slab (16 columns) at a time: two chunks per slab -------------------------
  f4 h3[DT];
#pragma unroll
  for (int nt = 0; nt < DT; ++nt) h3[nt] = *(const f4*)(P.bb2 + 16 * nt + 4 * g);
  for (int mt = 0; mt < IT; ++mt) {
    f4 act;
    {
      f4 frag[KT], b;
      st.fetch(frag, b);
      st.request_cols(P.w2, I, mt);  // A[n = 16t+c][m = 16mt+4g+r]
      PIN_ORDER();
      f4 h2[2] = {b, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[r & 1] = mfma16(frag[t][r], a[t][r], h2[r & 1]);
      const f4 pre = h2[0] + h2[1];
      f4 dact;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const PhiExp pe = phi_exp(pre[r]);
        act[r] = pre[r] * pe.phi;
        dact[r] = fmaf(pre[r] * kInvSqrt2Pi, pe.e, pe.phi);
      }
      if (ok) {
        *(f4*)(S.act + (size_t)row * I + 16 * mt + 4 * g) = act;
        if (S.gelu_grad) *(f4*)(S.gelu_grad + (size_t)row * I + 16 * mt + 4 * g) = dact;
      }
      PIN_ORDER();
      st.commit();
    }
    {
      f4 frag[KT], b;
      st.fetch(frag, b);
      if (mt + 1 < IT) st.request(P.w1, H, P.bb1, I, mt + 1, 0); else st.has_bias = false;
      PIN_ORDER();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt) h3[nt] = mfma16(frag[nt][r], act[r], h3[nt]);
      PIN_ORDER();
      st.commit();
    }
  }

  // ---- out = LayerNorm(dropout(h3) + a) -------------------------------------------------------------------------
  {
    f4 keep[DT], y[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) keep[t] = row_keep_scale(P.p2, P.keep2, P.seed2 + step, row, 4 * t + g, H);
    float mean, rstd;
    ln_forward_mem<DT>(h3, a, keep, P.g2, P.b2, g, P.eps2, y, mean, rstd);
    if (ok) {
#pragma unroll
      for (int t = 0; t < DT; ++t) {
        const size_t o = (size_t)row * H + 16 * t + 4 * g;
        *(f4*)(S.h3 + o) = h3[t];
        *(f4*)(S.out + o) = y[t];
      }
      if (g == 0) *(float2*)(S.st2 + 2 * (size_t)row) = float2{mean, rstd};
    }
  }
}

template <int H, int I, bool SAVED>
__global__ void __launch_bounds__(64 * NWV, 2) tail_staged_bwd_kernel(const acattn_tail_problem P, const acattn_tail_saved S,
                                                                      const acattn_tail_bwd_io IO, const float* __restrict__ ws) {
  constexpr int DT = H / 16, IT = I / 16;
  static_assert(DT == KT, "hidden 128");
  __shared__ f4 stage_lds[2 * STAGE_F4];
  WeightStage st;
  stage_init(st, stage_lds);
  const int c = st.c, g = st.g;
  const Rows<1> W = wg_tail_rows<1>(P, st.wave);
  const uint64_t step = P.seed_device ? *P.seed_device : 0ull;
  const int row = W.row[0];
  const bool ok = W.ok[0];
  // one partial row per ROW BLOCK, as in the per-wave kernels
  float* part = IO.dgb_part ? IO.dgb_part + (size_t)(blockIdx.x * NWV + st.wave) * 4 * H : nullptr;
  const bool write_part = (blockIdx.x * NWV + st.wave) * 16 < P.rows;
  const float *w1T = ws, *w2T = ws + (size_t)H * I, *wdT = ws + 2 * (size_t)H * I;
  if (SAVED) st.request(w2T, H, nullptr, I, 0, 0); else st.request(P.w1, H, P.bb1, I, 0, 0);

  // ---- through the second LayerNorm: d h3 (after the dropout), d a (residual share) -------------------------------
  f4 a[SAVED ? 1 : DT], dh3[DT], da[DT];
  {
    f4 z[DT], keep[DT], dy[DT], av[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const size_t o = (size_t)row * H + 16 * t + 4 * g;
      z[t] = *(const f4*)(S.h3 + o);
      av[t] = *(const f4*)(S.a + o);
      dy[t] = *(const f4*)(IO.d_out + o);
      keep[t] = row_keep_scale(P.p2, P.keep2, P.seed2 + step, row, 4 * t + g, H);
    }
    st.commit();
    const float2 st2 = *(const float2*)(S.st2 + 2 * (size_t)row);
    ln_backward_wide<DT>(z, av, keep, P.g2, st2.x, st2.y, dy, ok, write_part, c, g, da, part ? part + 2 * H : nullptr,
                         part ? part + 3 * H : nullptr);
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      dh3[t] = da[t] * keep[t];
      if (IO.d_h3 && ok) *(f4*)(IO.d_h3 + (size_t)row * H + 16 * t + 4 * g) = dh3[t];
      if constexpr (!SAVED) a[t] = av[t];
    }
  }

  // ---- d a += W1^T (gelu'(h2) * (W2^T d h3)), slab by slab --------------------------------------------------------------
  for (int mt = 0; mt < IT; ++mt) {
    f4 gp;  // gelu'(h2) of the slab
    if constexpr (SAVED) {
      gp = *(const f4*)(S.gelu_grad + (size_t)row * I + 16 * mt + 4 * g);
    } else {
      f4 frag[KT], b;
      st.fetch(frag, b);
      st.request(w2T, H, nullptr, I, mt, 0);
      PIN_ORDER();
      f4 h2[2] = {b, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[r & 1] = mfma16(frag[t][r], a[t][r], h2[r & 1]);
      const f4 pre = h2[0] + h2[1];
#pragma unroll
      for (int r = 0; r < 4; ++r) gp[r] = gelu_erf_grad(pre[r]);
      PIN_ORDER();
      st.commit();
    }
    f4 dh2;
    {
      f4 frag[KT], b;  // dense_2^T: A[m = 16mt+c][n = 16t+4g+r] = W2[n][m]
      st.fetch(frag, b);
      st.request_cols(w1T, I, mt);
      PIN_ORDER();
      f4 dact[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dact[r & 1] = mfma16(frag[t][r], dh3[t][r], dact[r & 1]);
      dh2 = (dact[0] + dact[1]) * gp;
      if (IO.d_h2 && ok) *(f4*)(IO.d_h2 + (size_t)row * I + 16 * mt + 4 * g) = dh2;
      PIN_ORDER();
      st.commit();
    }
    {
      f4 frag[KT], b;  // dense_1^T: A[k = 16nt+c][m = 16mt+4g+r] = W1[m][k]
      st.fetch(frag, b);
      if (mt + 1 < IT) {
        if (SAVED) st.request(w2T, H, nullptr, I, mt + 1, 0); else st.request(P.w1, H, P.bb1, I, mt + 1, 0);
      } else {
        st.request(wdT, H, nullptr, H, 0, 0);
      }
      PIN_ORDER();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt) da[nt] = mfma16(frag[nt][r], dh2[r], da[nt]);
      PIN_ORDER();
      st.commit();
    }
  }

  // ---- through the first LayerNorm: d h1, d x; then d ctx = d h1 . Wd ---------------------------------------------
  f4 dh1[DT];
  {
    f4 z[DT], res[DT], keep[DT], dz[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const size_t o = (size_t)row * H + 16 * t + 4 * g;
      z[t] = *(const f4*)(S.h1 + o);
      res[t] = *(const f4*)(P.x + (size_t)W.src[0] * H + 16 * t + 4 * g);
      keep[t] = row_keep_scale(P.p1, P.keep1, P.seed1 + step, row, 4 * t + g, H);
    }
    const float2 st1 = *(const float2*)(S.st1 + 2 * (size_t)row);
    ln_backward_wide<DT>(z, res, keep, P.g1, st1.x, st1.y, da, ok, write_part, c, g, dz, part, part ? part + H : nullptr);
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      const size_t o = (size_t)row * H + 16 * t + 4 * g;
      dh1[t] = dz[t] * keep[t];
      if (ok) {
        if (IO.d_x) store_grad(IO.d_x + (size_t)W.src[0] * H + 16 * t + 4 * g, dz[t], P.src_index != nullptr);
        if (IO.d_h1) *(f4*)(IO.d_h1 + o) = dh1[t];
      }
    }
  }
  // dense^T: A[k = 16nt+c][n = 16t+4g+r] = Wd[n][k] = WdT[16nt+c][16t+4g+r]  (walked also without d_ctx: the barriers)
  static_for<DT>([&](auto k) {
    constexpr int NT = decltype(k)::value;
    f4 frag[KT], b;
    st.fetch(frag, b);
    if constexpr (NT + 1 < DT) st.request(wdT, H, nullptr, H, NT + 1, 0);
    PIN_ORDER();
    f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r & 1] = mfma16(frag[t][r], dh1[t][r], acc[r & 1]);
    if (IO.d_ctx && ok) store_grad(IO.d_ctx + (size_t)W.src[0] * H + 16 * NT + 4 * g, acc[0] + acc[1], P.src_index != nullptr);
    PIN_ORDER();
    if constexpr (NT + 1 < DT) st.commit();
  });
}

// ---------------------------------------------------------------------------------------------------------------------
// Hidden 256 (BASELINE configs[4]; also runs at 128: ACATTN_TAIL_CHUNKED=1, the tests' cross-check): the staged chains
// with a tile's contraction split into H / 128 chunks and the LayerNorms STREAMED -- at 16 feature tiles per row four
// row-sized register arrays (value, residual, keep, cotangent) do not fit next to the chain's own two, so the forward
// normalises in place as the tiles of a product arrive and the backward walks its LayerNorm inputs twice (statistics,
// then gradients; the second walk hits L1 / L2).
// ---------------------------------------------------------------------------------------------------------------------
template <int H, int I>
__global__ void __launch_bounds__(64 * NWV, H > 128 ? 1 : (H < 128 ? ACATTN_TAIL64_WAVES : 2)) tail_chunked_fwd_kernel(const acattn_tail_problem P, const acattn_tail_saved S) {
  constexpr int KT = H >= 128 ? 8 : 4;  // fragments per staged chunk (shadows the file-level 8)
  constexpr int DT = H / 16, IT = I / 16, KC = DT / KT;
  constexpr float inv_h = 1.0f / H;
  using Stage = WeightStageT<KT>;
  __shared__ f4 stage_lds[2 * Stage::STAGE_F4];
  Stage st;
  stage_init(st, stage_lds);
  const int c = st.c, g = st.g;
  const Rows<1> W = wg_tail_rows<1>(P, st.wave);
  const uint64_t step = P.seed_device ? *P.seed_device : 0ull;
  const int row = W.row[0];
  const bool ok = W.ok[0];
  st.request(P.wd, H, P.bd, H, 0, 0);

  // ---- h1 = dense(ctx) + bias, tile by tile;  z = dropout(h1) + x collected in `a`;  a = LayerNorm(z) in place -----------
  f4 a[DT];
  {
    f4 cb[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t) cb[t] = *(const f4*)(P.ctx + (size_t)W.src[0] * H + 16 * t + 4 * g);
    st.commit();
    float sum = 0.f;
    static_for<DT>([&](auto k) {
      constexpr int NT = decltype(k)::value;
      const f4 res = *(const f4*)(P.x + (size_t)W.src[0] * H + 16 * NT + 4 * g);  // (in flight under the tile's MFMAs)
      f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, bias;
      static_for<KC>([&](auto q) {
        constexpr int C0 = decltype(q)::value;
        f4 frag[KT], b;
        st.fetch(frag, b);
        if constexpr (C0 == 0) bias = b;
        if constexpr (C0 + 1 < KC) st.request(P.wd, H, P.bd, H, NT, C0 + 1);
        else if constexpr (NT + 1 < DT) st.request(P.wd, H, P.bd, H, NT + 1, 0);
        else st.request(P.w1, H, P.bb1, I, 0, 0);
        PIN_ORDER();
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[r & 1] = mfma16(frag[t][r], cb[KT * C0 + t][r], acc[r & 1]);
        PIN_ORDER();
        st.commit();
      });
      const f4 h1 = (acc[0] + acc[1]) + bias;
      if (ok) *(f4*)(S.h1 + (size_t)row * H + 16 * NT + 4 * g) = h1;
      const f4 z = h1 * row_keep_scale(P.p1, P.keep1, P.seed1 + step, row, 4 * NT + g, H) + res;
      a[NT] = z;
      sum += hsum4(z);
    });
    const float mean = quad_sum(sum) * inv_h;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      a[t] = a[t] - mean;
      sq += hsum4(a[t] * a[t]);
    }
    const float rstd = __builtin_amdgcn_rsqf(quad_sum(sq) * inv_h + P.eps1);
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      a[t] = (a[t] * rstd) * *(const f4*)(P.g1 + 16 * t + 4 * g) + *(const f4*)(P.b1 + 16 * t + 4 * g);
      if (ok) *(f4*)(S.a + (size_t)row * H + 16 * t + 4 * g) = a[t];
    }
    if (ok && g == 0) *(float2*)(S.st1 + 2 * (size_t)row) = float2{mean, rstd};
  }

  // ---- h3 = dense_2(gelu(dense_1(a))), one inner slab at a time: KC chunks of dense_1's slab, KC of dense_2's columns ------
  f4 h3[DT];
#pragma unroll
  for (int nt = 0; nt < DT; ++nt) h3[nt] = *(const f4*)(P.bb2 + 16 * nt + 4 * g);
  for (int mt = 0; mt < IT; ++mt) {
    f4 h2[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, bias;
    static_for<KC>([&](auto q) {
      constexpr int C0 = decltype(q)::value;
      f4 frag[KT], b;
      st.fetch(frag, b);
      if constexpr (C0 == 0) bias = b;
      if constexpr (C0 + 1 < KC) st.request(P.w1, H, P.bb1, I, mt, C0 + 1); else st.request_cols(P.w2, I, mt);
      PIN_ORDER();
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[r & 1] = mfma16(frag[t][r], a[KT * C0 + t][r], h2[r & 1]);
      PIN_ORDER();
      st.commit();
    });
    const f4 pre = (h2[0] + h2[1]) + bias;
    f4 act, dact;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const PhiExp pe = phi_exp(pre[r]);
      act[r] = pre[r] * pe.phi;
      dact[r] = fmaf(pre[r] * kInvSqrt2Pi, pe.e, pe.phi);
    }
    if (ok) {
      *(f4*)(S.act + (size_t)row * I + 16 * mt + 4 * g) = act;
      if (S.gelu_grad) *(f4*)(S.gelu_grad + (size_t)row * I + 16 * mt + 4 * g) = dact;
    }
    static_for<KC>([&](auto q) {
      constexpr int C0 = decltype(q)::value;
      f4 frag[KT], b;
      st.fetch(frag, b);
      if constexpr (C0 + 1 < KC) {
        st.request_cols(P.w2 + (size_t)(16 * KT * (C0 + 1)) * I, I, mt);
      } else {
        if (mt + 1 < IT) st.request(P.w1, H, P.bb1, I, mt + 1, 0); else st.has_bias = false;
      }
      PIN_ORDER();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < KT; ++nt) h3[KT * C0 + nt] = mfma16(frag[nt][r], act[r], h3[KT * C0 + nt]);
      PIN_ORDER();
      st.commit();
    });
  }

  // ---- out = LayerNorm(dropout(h3) + a), in place in h3 ----------------------------------------------------------------
  {
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      if (ok) *(f4*)(S.h3 + (size_t)row * H + 16 * t + 4 * g) = h3[t];
      h3[t] = h3[t] * row_keep_scale(P.p2, P.keep2, P.seed2 + step, row, 4 * t + g, H) + a[t];
      sum += hsum4(h3[t]);
    }
    const float mean = quad_sum(sum) * inv_h;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      h3[t] = h3[t] - mean;
      sq += hsum4(h3[t] * h3[t]);
    }
    const float rstd = __builtin_amdgcn_rsqf(quad_sum(sq) * inv_h + P.eps2);
    if (ok) {
#pragma unroll
      for (int t = 0; t < DT; ++t)
        *(f4*)(S.out + (size_t)row * H + 16 * t + 4 * g) =
            (h3[t] * rstd) * *(const f4*)(P.g2 + 16 * t + 4 * g) + *(const f4*)(P.b2 + 16 * t + 4 * g);
      if (g == 0) *(float2*)(S.st2 + 2 * (size_t)row) = float2{mean, rstd};
    }
  }
}

// LayerNorm backward in two walks over its inputs in memory (z = pre-dropout value at `zp`, residual at `rp`; the cotangent
// comes from `dyp` or, when that is NULL, from the registers `dyr`): dz lands in `dz`
template <int DT, class Keep>
__device__ __forceinline__ void ln_backward_streamed(const float* zp, const float* rp, const float* dyp, const f4 (&dyr)[DT],
                                                     Keep&& keep_of, const float* gamma, float mean, float rstd, bool ok,
                                                     bool write_part, int c, int g, f4 (&dz)[DT], float* part_g, float* part_b) {
  constexpr float inv_h = 1.0f / (16 * DT);
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const f4 xh = ((*(const f4*)(zp + 16 * t + 4 * g) * keep_of(t) + *(const f4*)(rp + 16 * t + 4 * g)) - mean) * rstd;
    const f4 dy = dyp ? *(const f4*)(dyp + 16 * t + 4 * g) : dyr[t];
    const f4 gg = dy * *(const f4*)(gamma + 16 * t + 4 * g);
    m1 += hsum4(gg);
    m2 += hsum4(gg * xh);
    if (part_g) {
      f4 pg, pb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pg[r] = dpp_row_sum(ok ? dy[r] * xh[r] : 0.f);
        pb[r] = dpp_row_sum(ok ? dy[r] : 0.f);
      }
      if (c == 0 && write_part) {
        *(f4*)(part_g + 16 * t + 4 * g) = pg;
        *(f4*)(part_b + 16 * t + 4 * g) = pb;
      }
    }
  }
  m1 = quad_sum(m1) * inv_h;
  m2 = quad_sum(m2) * inv_h;
#pragma unroll
  for (int t = 0; t < DT; ++t) {
    const f4 xh = ((*(const f4*)(zp + 16 * t + 4 * g) * keep_of(t) + *(const f4*)(rp + 16 * t + 4 * g)) - mean) * rstd;
    const f4 dy = dyp ? *(const f4*)(dyp + 16 * t + 4 * g) : dyr[t];
    const f4 gg = dy * *(const f4*)(gamma + 16 * t + 4 * g);
    dz[t] = (gg - m1 - xh * m2) * rstd;
  }
}

template <int H, int I, bool SAVED>
__global__ void __launch_bounds__(64 * NWV, H > 128 ? 1 : (H < 128 ? ACATTN_TAIL64_WAVES : 2)) tail_chunked_bwd_kernel(const acattn_tail_problem P, const acattn_tail_saved S,
                                                                       const acattn_tail_bwd_io IO, const float* __restrict__ ws) {
  constexpr int KT = H >= 128 ? 8 : 4;  // fragments per staged chunk (shadows the file-level 8)
  constexpr int DT = H / 16, IT = I / 16, KC = DT / KT;
  using Stage = WeightStageT<KT>;
  __shared__ f4 stage_lds[2 * Stage::STAGE_F4];
  Stage st;
  stage_init(st, stage_lds);
  const int c = st.c, g = st.g;
  const Rows<1> W = wg_tail_rows<1>(P, st.wave);
  const uint64_t step = P.seed_device ? *P.seed_device : 0ull;
  const int row = W.row[0];
  const bool ok = W.ok[0];
  float* part = IO.dgb_part ? IO.dgb_part + (size_t)(blockIdx.x * NWV + st.wave) * 4 * H : nullptr;
  const bool write_part = (blockIdx.x * NWV + st.wave) * 16 < P.rows;
  const float *w1T = ws, *w2T = ws + (size_t)H * I, *wdT = ws + 2 * (size_t)H * I;
  if (SAVED) st.request(w2T, H, nullptr, I, 0, 0); else st.request(P.w1, H, P.bb1, I, 0, 0);

  // ---- through the second LayerNorm: d h3 (after the dropout), d a (residual share) -------------------------------
  f4 a[SAVED ? 1 : DT], dh3[DT], da[DT];
  {
    auto keep2 = [&](int t) { return row_keep_scale(P.p2, P.keep2, P.seed2 + step, row, 4 * t + g, H); };
    const float2 st2 = *(const float2*)(S.st2 + 2 * (size_t)row);
    st.commit();
    ln_backward_streamed<DT>(S.h3 + (size_t)row * H, S.a + (size_t)row * H, IO.d_out + (size_t)row * H, da, keep2, P.g2, st2.x,
                             st2.y, ok, write_part, c, g, da, part ? part + 2 * H : nullptr, part ? part + 3 * H : nullptr);
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      dh3[t] = da[t] * keep2(t);
      if (IO.d_h3 && ok) *(f4*)(IO.d_h3 + (size_t)row * H + 16 * t + 4 * g) = dh3[t];
      if constexpr (!SAVED) a[t] = *(const f4*)(S.a + (size_t)row * H + 16 * t + 4 * g);
    }
  }

  // ---- d a += W1^T (gelu'(h2) * (W2^T d h3)), slab by slab --------------------------------------------------------------
  for (int mt = 0; mt < IT; ++mt) {
    f4 gp;  // gelu'(h2) of the slab
    if constexpr (SAVED) {
      gp = *(const f4*)(S.gelu_grad + (size_t)row * I + 16 * mt + 4 * g);
    } else {
      f4 h2[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}}, bias;
      static_for<KC>([&](auto q) {
        constexpr int C0 = decltype(q)::value;
        f4 frag[KT], b;
        st.fetch(frag, b);
        if constexpr (C0 == 0) bias = b;
        if constexpr (C0 + 1 < KC) st.request(P.w1, H, P.bb1, I, mt, C0 + 1); else st.request(w2T, H, nullptr, I, mt, 0);
        PIN_ORDER();
#pragma unroll
        for (int t = 0; t < KT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) h2[r & 1] = mfma16(frag[t][r], a[SAVED ? 0 : KT * C0 + t][r], h2[r & 1]);
        PIN_ORDER();
        st.commit();
      });
      const f4 pre = (h2[0] + h2[1]) + bias;
#pragma unroll
      for (int r = 0; r < 4; ++r) gp[r] = gelu_erf_grad(pre[r]);
    }
    f4 dact[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
    static_for<KC>([&](auto q) {  // dense_2^T: A[m = 16mt+c][n = 16t+4g+r] = W2[n][m]
      constexpr int C0 = decltype(q)::value;
      f4 frag[KT], b;
      st.fetch(frag, b);
      if constexpr (C0 + 1 < KC) st.request(w2T, H, nullptr, I, mt, C0 + 1); else st.request_cols(w1T, I, mt);
      PIN_ORDER();
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) dact[r & 1] = mfma16(frag[t][r], dh3[KT * C0 + t][r], dact[r & 1]);
      PIN_ORDER();
      st.commit();
    });
    const f4 dh2 = (dact[0] + dact[1]) * gp;
    if (IO.d_h2 && ok) *(f4*)(IO.d_h2 + (size_t)row * I + 16 * mt + 4 * g) = dh2;
    static_for<KC>([&](auto q) {  // dense_1^T: A[k = 16nt+c][m = 16mt+4g+r] = W1[m][k]
      constexpr int C0 = decltype(q)::value;
      f4 frag[KT], b;
      st.fetch(frag, b);
      if constexpr (C0 + 1 < KC) {
        st.request_cols(w1T + (size_t)(16 * KT * (C0 + 1)) * I, I, mt);
      } else {
        if (mt + 1 < IT) {
          if (SAVED) st.request(w2T, H, nullptr, I, mt + 1, 0); else st.request(P.w1, H, P.bb1, I, mt + 1, 0);
        } else {
          st.request(wdT, H, nullptr, H, 0, 0);
        }
      }
      PIN_ORDER();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < KT; ++nt) da[KT * C0 + nt] = mfma16(frag[nt][r], dh2[r], da[KT * C0 + nt]);
      PIN_ORDER();
      st.commit();
    });
  }

  // ---- through the first LayerNorm: d h1 (in da's registers), d x; then d ctx = d h1 . Wd --------------------------------
  {
    auto keep1 = [&](int t) { return row_keep_scale(P.p1, P.keep1, P.seed1 + step, row, 4 * t + g, H); };
    const float2 st1 = *(const float2*)(S.st1 + 2 * (size_t)row);
    f4 dz[DT];
    ln_backward_streamed<DT>(S.h1 + (size_t)row * H, P.x + (size_t)W.src[0] * H, nullptr, da, keep1, P.g1, st1.x, st1.y, ok,
                             write_part, c, g, dz, part, part ? part + H : nullptr);
#pragma unroll
    for (int t = 0; t < DT; ++t) {
      da[t] = dz[t] * keep1(t);  // d h1
      if (ok) {
        if (IO.d_x) store_grad(IO.d_x + (size_t)W.src[0] * H + 16 * t + 4 * g, dz[t], P.src_index != nullptr);
        if (IO.d_h1) *(f4*)(IO.d_h1 + (size_t)row * H + 16 * t + 4 * g) = da[t];
      }
    }
  }
  // dense^T: A[k = 16nt+c][n = 16t+4g+r] = Wd[n][k] = WdT[16nt+c][16t+4g+r]  (walked also without d_ctx: the barriers)
  static_for<DT>([&](auto k) {
    constexpr int NT = decltype(k)::value;
    f4 acc[2] = {f4{0.f, 0.f, 0.f, 0.f}, f4{0.f, 0.f, 0.f, 0.f}};
    static_for<KC>([&](auto q) {
      constexpr int C0 = decltype(q)::value;
      constexpr bool last = NT + 1 == DT && C0 + 1 == KC;
      f4 frag[KT], b;
      st.fetch(frag, b);
      if constexpr (C0 + 1 < KC) st.request(wdT, H, nullptr, H, NT, C0 + 1);
      else if constexpr (NT + 1 < DT) st.request(wdT, H, nullptr, H, NT + 1, 0);
      PIN_ORDER();
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r & 1] = mfma16(frag[t][r], da[KT * C0 + t][r], acc[r & 1]);
      PIN_ORDER();
      if constexpr (!last) st.commit();
    });
    if (IO.d_ctx && ok) store_grad(IO.d_ctx + (size_t)W.src[0] * H + 16 * NT + 4 * g, acc[0] + acc[1], P.src_index != nullptr);
  });
}

template <int H, int I>
int launch_wide_fwd(const acattn_tail_problem& p, const acattn_tail_saved& s, hipStream_t stream) {
  const int blocks = (p.rows + 15) / 16;
  static const bool per_wave = getenv("ACATTN_TAIL_PER_WAVE") != nullptr;  // measurement: the per-wave weight stream
  static const bool chunked = getenv("ACATTN_TAIL_CHUNKED") != nullptr;    // cross-check: the hidden-256 form at 128
  if (H > 128 || chunked) {
    hipLaunchKernelGGL((tail_chunked_fwd_kernel<H, I>), dim3((blocks + NWV - 1) / NWV), dim3(64 * NWV), 0, stream, p, s);
    return (int)hipGetLastError();
  }
  if constexpr (H <= 128) {
  if (wide_split(p.rows))
    hipLaunchKernelGGL((tail_wide_fwd_kernel<H, I, 4>), dim3(blocks), dim3(256), 0, stream, p, s);
  else if (per_wave)
    hipLaunchKernelGGL((tail_wide_fwd_kernel<H, I, 1>), dim3(blocks), dim3(64), 0, stream, p, s);
  else
    hipLaunchKernelGGL((tail_staged_fwd_kernel<H, I>), dim3((blocks + NWV - 1) / NWV), dim3(64 * NWV), 0, stream, p, s);
  }
  return (int)hipGetLastError();
}

template <int H, int I>
int launch_wide_bwd(const acattn_tail_problem& p, const acattn_tail_saved& s, const acattn_tail_bwd_io& io, hipStream_t stream) {
  if (!io.workspace) {
    acattn_set_error("layer tail backward at hidden 128 / 256 needs acattn_tail_bwd_io.workspace");
    return -1;
  }
  float* ws = (float*)io.workspace;
  hipLaunchKernelGGL(tail_transpose_kernel, dim3(I / 16, I / 16, 3), dim3(256), 0, stream, p.w1, p.w2, p.wd, H, I, ws);
  const int blocks = (p.rows + 15) / 16;
  const bool split = wide_split(p.rows);
  static const bool per_wave = getenv("ACATTN_TAIL_PER_WAVE") != nullptr;  // measurement: the per-wave weight stream
  static const bool chunked = getenv("ACATTN_TAIL_CHUNKED") != nullptr;    // cross-check: the hidden-256 form at 128
  if (H > 128 || chunked) {
    const int wgs = (blocks + NWV - 1) / NWV;
    if (s.gelu_grad)
      hipLaunchKernelGGL((tail_chunked_bwd_kernel<H, I, true>), dim3(wgs), dim3(64 * NWV), 0, stream, p, s, io, (const float*)ws);
    else
      hipLaunchKernelGGL((tail_chunked_bwd_kernel<H, I, false>), dim3(wgs), dim3(64 * NWV), 0, stream, p, s, io, (const float*)ws);
    return (int)hipGetLastError();
  }
  if constexpr (H <= 128) {
  if (!split && !per_wave) {
    const int wgs = (blocks + NWV - 1) / NWV;
    if (s.gelu_grad)
      hipLaunchKernelGGL((tail_staged_bwd_kernel<H, I, true>), dim3(wgs), dim3(64 * NWV), 0, stream, p, s, io, (const float*)ws);
    else
      hipLaunchKernelGGL((tail_staged_bwd_kernel<H, I, false>), dim3(wgs), dim3(64 * NWV), 0, stream, p, s, io, (const float*)ws);
    return (int)hipGetLastError();
  }
  if (s.gelu_grad) {
    if (split)
      hipLaunchKernelGGL((tail_wide_bwd_kernel<H, I, 4, true>), dim3(blocks), dim3(256), 0, stream, p, s, io, (const float*)ws);
    else
      hipLaunchKernelGGL((tail_wide_bwd_kernel<H, I, 1, true>), dim3(blocks), dim3(64), 0, stream, p, s, io, (const float*)ws);
  } else {
    if (split)
      hipLaunchKernelGGL((tail_wide_bwd_kernel<H, I, 4, false>), dim3(blocks), dim3(256), 0, stream, p, s, io, (const float*)ws);
    else
      hipLaunchKernelGGL((tail_wide_bwd_kernel<H, I, 1, false>), dim3(blocks), dim3(64), 0, stream, p, s, io, (const float*)ws);
  }
  }
  return (int)hipGetLastError();
}

template <int H, int I>
int launch_fwd(const acattn_tail_problem& p, const acattn_tail_saved& s, hipStream_t stream) {
  if (staged64(p.rows)) {
    const int wgs = (p.rows + 16 * NWV - 1) / (16 * NWV);
    hipLaunchKernelGGL((tail_chunked_fwd_kernel<H, I>), dim3(wgs), dim3(64 * NWV), 0, stream, p, s);
    return (int)hipGetLastError();
  }
  const int nb = rows_per_wave(p.rows) / 16;
  const int blocks = (p.rows + 16 * nb - 1) / (16 * nb);
  if (nb == 1 && split_slabs(p.rows)) {
    hipLaunchKernelGGL((tail_fwd_kernel<H, I, 1, 4>), dim3(blocks), dim3(256), 0, stream, p, s);
    return (int)hipGetLastError();
  }
  if (nb == 2)
    hipLaunchKernelGGL((tail_fwd_kernel<H, I, 2>), dim3(blocks), dim3(64), 0, stream, p, s);
  else
    hipLaunchKernelGGL((tail_fwd_kernel<H, I, 1>), dim3(blocks), dim3(64), 0, stream, p, s);
  return (int)hipGetLastError();
}

template <int H, int I>
int launch_bwd(const acattn_tail_problem& p, const acattn_tail_saved& s, const acattn_tail_bwd_io& io, hipStream_t stream) {
  if (staged64(p.rows)) {
    if (!io.workspace) {
      acattn_set_error("layer tail backward: this row count needs acattn_tail_bwd_io.workspace (acattn_layer_tail_bwd_workspace_bytes)");
      return -1;
    }
    float* ws = (float*)io.workspace;
    hipLaunchKernelGGL(tail_transpose_kernel, dim3(I / 16, I / 16, 3), dim3(256), 0, stream, p.w1, p.w2, p.wd, H, I, ws);
    const int wgs = (p.rows + 16 * NWV - 1) / (16 * NWV);
    if (s.gelu_grad)
      hipLaunchKernelGGL((tail_chunked_bwd_kernel<H, I, true>), dim3(wgs), dim3(64 * NWV), 0, stream, p, s, io, (const float*)ws);
    else
      hipLaunchKernelGGL((tail_chunked_bwd_kernel<H, I, false>), dim3(wgs), dim3(64 * NWV), 0, stream, p, s, io, (const float*)ws);
    return (int)hipGetLastError();
  }
  const int nb = rows_per_wave(p.rows) / 16;
  const int blocks = acattn_tail_bwd_partial_rows(p.rows);
  if (nb == 1 && split_slabs(p.rows)) {
    hipLaunchKernelGGL((tail_bwd_kernel<H, I, 1, 4>), dim3(blocks), dim3(256), 0, stream, p, s, io);
    return (int)hipGetLastError();
  }
  if (nb == 2)
    hipLaunchKernelGGL((tail_bwd_kernel<H, I, 2>), dim3(blocks), dim3(64), 0, stream, p, s, io);
  else
    hipLaunchKernelGGL((tail_bwd_kernel<H, I, 1>), dim3(blocks), dim3(64), 0, stream, p, s, io);
  return (int)hipGetLastError();
}

}  // namespace

bool acattn_tail_supported(int H, int I) {
  return (H == 64 && (I == 256 || I == 128)) || (H == 128 && (I == 512 || I == 256)) || (H == 256 && I == 1024);
}

int acattn_tail_bwd_partial_rows(int rows) { return (rows + rows_per_wave(rows) - 1) / rows_per_wave(rows); }
int acattn_tail_bwd_partial_rows_h(int rows, int H) {
  return (H > 64 || staged64(rows)) ? (rows + 15) / 16 : acattn_tail_bwd_partial_rows(rows);
}
// (hidden 64 needs it for the staged form only, i.e. above the slab-split row count; asking for it always is simpler)
int64_t acattn_tail_bwd_ws_bytes(int H, int I) { return ((int64_t)2 * H * I + (int64_t)H * H) * (int64_t)sizeof(float); }

int acattn_select_tail_nb(int nb) {
  const int prev = g_tail_nb;
  g_tail_nb = nb;
  return prev;
}

int acattn_launch_tail_fwd(const acattn_tail_problem& p, const acattn_tail_saved& s, hipStream_t stream) {
  if (p.H == 64 && p.I == 256) return launch_fwd<64, 256>(p, s, stream);
  if (p.H == 64 && p.I == 128) return launch_fwd<64, 128>(p, s, stream);
  if (p.H == 128 && p.I == 512) return launch_wide_fwd<128, 512>(p, s, stream);
  if (p.H == 128 && p.I == 256) return launch_wide_fwd<128, 256>(p, s, stream);
  if (p.H == 256 && p.I == 1024) return launch_wide_fwd<256, 1024>(p, s, stream);
  acattn_set_error("layer tail: unsupported (hidden_size, inner_size)");
  return -1;
}

int acattn_launch_tail_bwd(const acattn_tail_problem& p, const acattn_tail_saved& s, const acattn_tail_bwd_io& io,
                           hipStream_t stream) {
  if (p.H == 64 && p.I == 256) return launch_bwd<64, 256>(p, s, io, stream);
  if (p.H == 64 && p.I == 128) return launch_bwd<64, 128>(p, s, io, stream);
  if (p.H == 128 && p.I == 512) return launch_wide_bwd<128, 512>(p, s, io, stream);
  if (p.H == 128 && p.I == 256) return launch_wide_bwd<128, 256>(p, s, io, stream);
  if (p.H == 256 && p.I == 1024) return launch_wide_bwd<256, 1024>(p, s, io, stream);
  acattn_set_error("layer tail: unsupported (hidden_size, inner_size)");
  return -1;
}
