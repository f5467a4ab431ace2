// Backward of the fused calibrated attention, streaming form (any L <= 208): two kernels, no score tile ever kept
// across key tiles, no LDS staging, no float atomics on activations.
//
// Why.  The row-resident backward kernels (acattn_bwd.hip, acattn_bwd_fast.hip) hold a query block's whole row of
// P, M and their cotangents in registers: at L = 200 that is 512 registers plus 3.5 KB of scratch per lane, one wave
// per SIMD, and 7 ms per launch at B = 512 (the forward: 0.36 ms).  Every probability tensor of the layer can be
// rebuilt tile by tile from the five log-normalisers the forward saved, and the five row scalars the chained
// soft-max backward needs
//     da = <A_p, dA_p>   dc = <A_w, dA_w>   r1 = <A_c, dA_c>   sP = <P, dP>   sM = <M, dM>
// are AFFINE in each other with tile-local coefficients (dA_c, dP, dM are linear in dc, r1, da), so ONE sweep over
// the key tiles accumulates twelve sums from which all five follow.  That gives
//
//   row kernel   one wave per (sequence, head, 16-row query block):
//                sweep 1 -> the five row scalars (kept in a small workspace for the key kernel);
//                sweep 2 -> dS, dSa tile by tile -> dq, dqa (MFMA, accumulated in registers), the gate-logit
//                partials, the query halves of the calibrator parameter gradients;
//   key kernel   one wave per (sequence, head, 16-key tile): sweeps the query blocks that see the tile, rebuilds
//                the same tiles, turns them through a 5 KB LDS scratch and accumulates dK, dKa, dV in registers
//                (MFMA), plus the key halves of the calibrator parameter gradients.
//
// Three recomputations of the forward's elementwise work instead of one, but at 2-3 waves per SIMD with nothing
// spilled, and a grid of thousands of independent waves.  Same mathematics as acattn_bwd_fast.hip (which stays the
// L <= 64 path: there the row fits comfortably and one recomputation wins).
// Reference: recbole/model/layers.py:657-742, 883-951 (what is differentiated).
#include <stdlib.h>

#include <algorithm>

#include "acattn_common.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float and_bits(float x, int m) { return __uint_as_float(__float_as_uint(x) & (uint32_t)m); }
__device__ __forceinline__ float hsum(const f4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }
__device__ __forceinline__ float row16_sum(float v) {  // over the 16 lanes of a DPP row (same g, all c)
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  return v;
}

// head sizes from this one up run one wave per SIMD (probe: -DACATTN_BWD_ONE_WAVE_DH=32)
#ifndef ACATTN_BWD_ONE_WAVE_DH
#define ACATTN_BWD_ONE_WAVE_DH 64
#endif
constexpr int NSC = 8;  // floats per row in the workspace: da, dc, r1, sP, sM, (3 spare)

// Everything a lane knows about ITS query row (row i0 + c of one query block) that does not depend on the key tile.
template <int DH>
struct Row {
  static constexpr int KS = DH / 4;
  float qf[KS], qaf[KS];    // B operands of S^T = K.Q^T and Sa^T = Ka.Qa^T
  float gaf[KS], gcf[KS];   // d_ctx_attacked / d_ctx_calibrated of the row (B operands of dA^T = V.dctx^T)
  float gcf2[KS];           // [r4] d_ctx_calibrated of the SECOND cotangent set (acattn_bwd_io.d_ctx_calibrated2), TWO only
  float ao2, ad;            // query halves of the order (exp2 domain) / distance affines
  float lx2, ly2, lu2, lv2, lw2;  // the forward's log-normalisers, exp2 domain, dead-row shift removed
  int i;
  bool row_ok, dead;
  uint32_t rng_row;
};

// Constants of a launch (wave-uniform).
struct Consts {
  float inv_sqrt, scale2, nc2, s2, sc, keep_scale, p_drop;
  bool has_drop, causal;
  int gate_is_prob;  // acattn_problem.gate_is_prob
  RngKey rkey;
};

// One (query block, key tile) pair, forward part: every probability tensor of the layer, in the D layout of
// S^T = K.Q^T (lane (c, g): query row c, keys 16 t + 4 g + r).
struct Tile {
  f4 Pt, Mt;        // softmax(x), softmax(y) before dropout
  f4 P, M;          // after dropout (M is the layer's attack mask)
  f4 Ap, Ac, Aw;    // perturbed, calibrated, combined attention
  f4 gt, ex1, nz;   // gate, exp(1 - M), noise
  f4 dAp, dAw;      // cotangents of A_p and A_w (dctx . V^T)
  f4 dAw2;          // [r4] d A_w of the second cotangent set (TWO only)
  f4 pr, val, df;   // spatial calibrator: sigmoid(o), its log argument, distance residual
  uint32_t ka, km;  // dropout keep bits
  int eb[4];        // -1 where the key may receive probability mass
};

// Everything of tile_forward that follows the four products (raw scores S, Sa and the cotangents dA_p, dA_w of the
// lane's 4 keys): RowT supplies the row's scalars (Row<DH>, or RowScalars of the one-row kernel).
// DCA = false [r4]: the launch has NO cotangent of the attacked context (every layer but the last: its attacked branch feeds
// nothing, layers.py:1112).  Then d A_p = 0, so A_p itself, its argument and the Gaussian noise that only enters through it
// (layers.py:917-919) are never needed: the Box-Muller draws -- a sixth of the tile's arithmetic -- fold away, one of the
// four products (d A_p = d ctx_att . V^T) and its d V term go, and the row scalars da, the sums that carry it, vanish.
template <class RowT, bool DCA = true>
__device__ __forceinline__ void tile_elementwise(const RowT& R, const Consts& K, const f4 aS, const f4 aM, const f4 aP,
                                                 const f4 aW, const f4 co4, const f4 cd4, const f4 gl, const int t,
                                                 const int g, const uint32_t eb4, const uint32_t ab4,
                                                 const bool order_select, Tile& T) {
  T.dAp = aP;
  T.dAw = aW;
  const f4 ea = co4 + R.ao2;
  const int d0 = R.i - (16 * t + 4 * g);
  f4 lt4, mk4, lg;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    T.eb[r] = sbit(eb4, r);
    const float pr = fast_rcp(1.0f + ex2(ea[r]));
    T.pr[r] = pr;
    float val = 1.0f - pr;
    if (order_select) val = sbit(ab4, r) ? pr : val;
    T.val[r] = val;
    lt4[r] = __builtin_amdgcn_logf((float)(abs(d0 - r) + 1)) * kLn2;
    mk4[r] = and_bits(ACATTN_MASK_FILL * kLog2e, ~T.eb[r]);
    lg[r] = __builtin_amdgcn_logf(val + ACATTN_LOG_EPS);
  }
  T.df = lt4 - (cd4 + R.ad);
  f4 x = aS * K.scale2 + mk4;
  x = lg * K.inv_sqrt + x;
  x = (T.df * T.df) * K.nc2 + x;
  const f4 y = aM * K.scale2 + mk4;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    T.Pt[r] = ex2(x[r] - R.lx2);
    T.Mt[r] = ex2(y[r] - R.ly2);
  }
  const RngGroup rg = rng_group(K.rkey, R.rng_row, (uint32_t)(4 * t + g), K.p_drop);
  T.nz = rg.n;
  T.ka = K.has_drop ? rg.keep_after : 0xFu;
  T.km = K.has_drop ? rg.keep_mask : 0xFu;
  T.P = T.Pt * K.keep_scale;
  T.M = T.Mt * K.keep_scale;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    T.P[r] = keep_and(T.P[r], T.ka, r);
    T.M[r] = keep_and(T.M[r], T.km, r);
  }
  const f4 a1 = T.M * (-kLog2e) + kLog2e;
  if constexpr (DCA) {
    const f4 au = (T.P * T.M + T.nz * (1.0f - T.M)) * kLog2e - R.lu2;  // layers.py:918-919
#pragma unroll
    for (int r = 0; r < 4; ++r) T.Ap[r] = and_bits(ex2(au[r]), T.eb[r]);
  } else {
    T.Ap = f4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) T.ex1[r] = ex2(a1[r]);
  const f4 av = (T.P * T.ex1) * kLog2e - R.lv2;  // layers.py:920-921
#pragma unroll
  for (int r = 0; r < 4; ++r) T.Ac[r] = and_bits(ex2(av[r]), T.eb[r]);
  T.gt = gate_value(gl, K.gate_is_prob);  // layers.py:887
  const f4 aw = (T.gt * (T.P - T.Ac) + T.Ac) * kLog2e - R.lw2;  // layers.py:888, 925
#pragma unroll
  for (int r = 0; r < 4; ++r) T.Aw[r] = and_bits(ex2(aw[r]), T.eb[r]);
}

template <int DH, bool DCA = true, bool TWO = false>
__device__ __forceinline__ void tile_forward(const Row<DH>& R, const Consts& K, const f4 (&k4)[DH / 16], const f4 (&ka4)[DH / 16],
                                             const f4 (&v4)[DH / 16], const f4 co4, const f4 cd4, const f4 gl, const int t,
                                             const int g, const uint32_t eb4, const uint32_t ab4, const bool order_select,
                                             Tile& T) {
  constexpr int KS = DH / 4;
  f4 aS = {0.f, 0.f, 0.f, 0.f}, aM = aS, aP = aS, aW = aS, aW2 = aS;
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      aS = mfma16(k4[s4][e], R.qf[4 * s4 + e], aS);
      aM = mfma16(ka4[s4][e], R.qaf[4 * s4 + e], aM);
      if constexpr (DCA) aP = mfma16(v4[s4][e], R.gaf[4 * s4 + e], aP);
      aW = mfma16(v4[s4][e], R.gcf[4 * s4 + e], aW);
      if constexpr (TWO) aW2 = mfma16(v4[s4][e], R.gcf2[4 * s4 + e], aW2);
    }
  }
  T.dAw2 = aW2;
  tile_elementwise<Row<DH>, DCA>(R, K, aS, aM, aP, aW, co4, cd4, gl, t, g, eb4, ab4, order_select, T);
}

// Backward part once the row scalars are known: dS, dSa (scores), the gate-logit gradient, d o and d d of the two
// spatial affines, and the scalar's partial.
template <bool DCA = true>
__device__ __forceinline__ void tile_backward(const Tile& T, const Consts& K, const float da, const float dc, const float r1,
                                              const float sP, const float sM, const f4 dMout, const int i, const int j0,
                                              f4& dS, f4& dSa, f4& dgl, f4& d_o, f4& d_d, float& dsc) {
  f4 du = {0.f, 0.f, 0.f, 0.f};
  if constexpr (DCA) du = T.Ap * (T.dAp - da);
  const f4 dw = T.Aw * (T.dAw - dc);  // = d A_g
  dgl = dw * (T.P - T.Ac) * (T.gt * (1.0f - T.gt));
  const f4 dac = (1.0f - T.gt) * dw;
  const f4 dv = T.Ac * (dac - r1);
  f4 dP = T.gt * dw + dv * T.ex1;
  f4 dM = dMout - dv * (T.P * T.ex1);
  if constexpr (DCA) {
    dP += du * T.M;
    dM += du * (T.P - T.nz);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {  // through the dropouts: kept entries are scaled, dropped ones carry nothing
    dP[r] = keep_and(dP[r] * K.keep_scale, T.ka, r);
    dM[r] = keep_and(dM[r] * K.keep_scale, T.km, r);
  }
  dS = (T.Pt * (dP - sP)) * K.inv_sqrt;
  dSa = (T.Mt * (dM - sM)) * K.inv_sqrt;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float sgn = (j0 + r > i) ? 1.0f : -1.0f;  // d val / d sigmoid: +1 for keys after the query, -1 otherwise
    d_o[r] = dS[r] * (sgn * T.pr[r] * (1.0f - T.pr[r])) * fast_rcp(T.val[r] + ACATTN_LOG_EPS);
  }
  d_d = dS * (T.df * K.s2);
  dsc = hsum(dS * (T.df * T.df)) * (-K.sc);
}

// [r4] A (query block, key tile) pair whose rows carry NO context cotangent (acattn_bwd_io.read_rows / active_qblocks: in the
// last layer every block but the one with the read position): what reaches it is the mask's cotangent alone -- the
// penalty's d M = 2 d_pen (M - 1) (acsasrec.py:131-137) and / or a dense d M -- and d M only flows back through
// M = dropout(softmax(Sa / sqrt(dh) + mask)) (layers.py:664-672): du = dv = dw = 0 in tile_backward, so dS = 0, the gate
// and the spatial calibrator receive nothing, and of the tile only Mt / M have to be rebuilt (one product, four
// exponentials, the keep bits) instead of all seven probability tensors and the random numbers.
// `Mt`: softmax(y) before dropout; `dM`: the cotangent of Mt's argument side, i.e. keep(d M_out * keep_scale).
template <class RowT>
__device__ __forceinline__ void mask_tile(const RowT& R, const Consts& K, const f4 aM, const int t, const int g, const uint32_t eb4,
                                          const f4 dMext, const float dpen2, f4& Mt, f4& dM) {
  f4 mk4;
#pragma unroll
  for (int r = 0; r < 4; ++r) mk4[r] = and_bits(ACATTN_MASK_FILL * kLog2e, ~sbit(eb4, r));
  const f4 y = aM * K.scale2 + mk4;
#pragma unroll
  for (int r = 0; r < 4; ++r) Mt[r] = ex2(y[r] - R.ly2);
  uint32_t km = 0xFu;
  if (K.has_drop) km = rng_group(K.rkey, R.rng_row, (uint32_t)(4 * t + g), K.p_drop).keep_mask;  // (the normals fold away)
  f4 M = Mt * K.keep_scale;
#pragma unroll
  for (int r = 0; r < 4; ++r) M[r] = keep_and(M[r], km, r);
  dM = dMext + (M - 1.0f) * dpen2;
#pragma unroll
  for (int r = 0; r < 4; ++r) dM[r] = keep_and(dM[r] * K.keep_scale, km, r);
}

// [r4] The SECOND cotangent set of a launch (the single-pass combined backward, ac_tsr_amd/combined.py): the attacked loss's
// cotangents of a layer that has no attack transform upstream -- d ctx_calibrated2 (what the layer above hands down) and the
// mask penalty's d_pen2; no attacked context, and of its gradients only dqa, dka are wanted (trainer.py:678-684).  It shares
// the rebuilt tile with the first set: d A_p = 0 (du = 0), so dS is not needed and dSa follows from dc2, r1_2, sM2 alone.
__device__ __forceinline__ f4 second_set_dsa(const Tile& T, const Consts& K, const float dc2, const float r1_2, const float sM2,
                                             const f4 dMout2) {
  const f4 dw = T.Aw * (T.dAw2 - dc2);
  const f4 dac = (1.0f - T.gt) * dw;
  const f4 dv = T.Ac * (dac - r1_2);
  f4 dM = dMout2 - dv * (T.P * T.ex1);
#pragma unroll
  for (int r = 0; r < 4; ++r) dM[r] = keep_and(dM[r] * K.keep_scale, T.km, r);
  return (T.Mt * (dM - sM2)) * K.inv_sqrt;
}

// 4 floats of a [.., L] row at key offset j0 (rows with L % 4 != 0 are only dword aligned; the last group is ragged)
__device__ __forceinline__ f4 load_seg(const float* row, int j0, int L, bool ok) {
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (ok && j0 < L) {
    if (j0 + 3 < L) {
      v = *(const f4u*)(row + j0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j0 + r < L) v[r] = row[j0 + r];
    }
  }
  return v;
}
__device__ __forceinline__ void store_seg(float* row, int j0, int L, bool ok, const f4 val) {
  if (!ok || j0 >= L) return;
  if (j0 + 3 < L) {
    *(f4u*)(row + j0) = val;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (j0 + r < L) row[j0 + r] = val[r];
  }
}

// Wave-level description of a sequence's key flags (up to 208 keys: four 64-bit ballots).
struct KeyFlags {
  unsigned long long vk[4];
  int first_valid, nt_valid;
  bool any_valid;
};
__device__ __forceinline__ KeyFlags load_key_flags(const uint8_t* key_valid, size_t rowbase, int L, int nT, int lane) {
  KeyFlags F;
  F.any_valid = false;
  F.first_valid = L;
  int last_valid = -1;
#pragma unroll
  for (int q = 3; q >= 0; --q) {
    const uint8_t b = (64 * q + lane < L) ? key_valid[rowbase + 64 * q + lane] : (uint8_t)0;
    F.vk[q] = __ballot(b != 0);
    if (F.vk[q]) {
      F.any_valid = true;
      F.first_valid = 64 * q + __ffsll((long long)F.vk[q]) - 1;
      if (last_valid < 0) last_valid = 64 * q + 63 - __clzll((long long)F.vk[q]);
    }
  }
  F.nt_valid = F.any_valid ? (last_valid >> 4) + 1 : nT;
  return F;
}
// the lane's 4 mask bits of key tile t for query row i: `eb` = may receive probability mass, `ab` = key after the query
__device__ __forceinline__ void tile_bits(const KeyFlags& F, int t, int g, int i, int L, bool causal, bool row_ok, bool dead,
                                          uint32_t& eb, uint32_t& ab) {
  const uint32_t vn = (uint32_t)(F.vk[t >> 2] >> (16 * (t & 3) + 4 * g)) & 0xFu;
  const int d0 = i - 16 * t - 4 * g;
  const uint32_t cn = d0 >= 3 ? 0xFu : (d0 < 0 ? 0u : (2u << d0) - 1u);
  const int n_in = min(max(L - 16 * t - 4 * g, 0), 4);
  const uint32_t inr = (1u << n_in) - 1u;
  ab = ~cn & 0xFu;
  eb = dead ? inr : (causal ? (vn & cn) : vn);
  if (!row_ok) eb = 0u;  // rows past L: everything masked, every probability exactly 0
}

// Query-row state of block qb for lane c.
template <int DH>
__device__ __forceinline__ void load_row(const acattn_problem& P, const acattn_bwd_io& IO, const KeyFlags& F, size_t rowbase,
                                         size_t bh, int hoff, int qb, int c, int g, Row<DH>& R) {
  constexpr int KS = DH / 4;
  const int L = P.L, H = P.H;
  R.i = 16 * qb + c;
  R.row_ok = R.i < L;
  const size_t off = (rowbase + (R.row_ok ? R.i : 0)) * H + hoff + KS * g;
  float ao = 0.f, adv = 0.f;
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    const f4 tq = *(const f4*)(P.q + off + 4 * s4), ta = *(const f4*)(P.qa + off + 4 * s4);
    f4 ga = {0.f, 0.f, 0.f, 0.f}, gc = ga, gc2 = ga;
    if (IO.d_ctx_attacked) ga = *(const f4*)(IO.d_ctx_attacked + off + 4 * s4);
    if (IO.d_ctx_calibrated) gc = *(const f4*)(IO.d_ctx_calibrated + off + 4 * s4);
    if (IO.d_ctx_calibrated2) gc2 = *(const f4*)(IO.d_ctx_calibrated2 + off + 4 * s4);
    const f4 a = *(const f4*)(P.w_order + KS * g + 4 * s4), d = *(const f4*)(P.w_dist + KS * g + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      R.qf[4 * s4 + e] = R.row_ok ? tq[e] : 0.f;
      R.qaf[4 * s4 + e] = R.row_ok ? ta[e] : 0.f;
      R.gaf[4 * s4 + e] = R.row_ok ? ga[e] : 0.f;
      R.gcf[4 * s4 + e] = R.row_ok ? gc[e] : 0.f;
      R.gcf2[4 * s4 + e] = R.row_ok ? gc2[e] : 0.f;
      ao += R.qf[4 * s4 + e] * a[e];
      adv += R.qf[4 * s4 + e] * d[e];
    }
  }
  ao = quad_sum(ao) + P.b_order[0];
  R.ad = quad_sum(adv) + P.b_dist[0];
  R.ao2 = -kLog2e * ao;
  // a row without any allowed key (left padding) spreads over every key < L; the forward added its -10000 shift to
  // the stored normalisers (acattn_fwd_body.inc), which is taken out again here
  R.dead = P.causal ? F.first_valid > R.i : !F.any_valid;
  const float sh = R.dead ? ACATTN_MASK_FILL : 0.f;
  const float* sp = IO.row_stats + (bh * L + (R.row_ok ? R.i : 0)) * ACATTN_NSTAT;
  const f4 s0 = *(const f4*)sp;
  const float s4v = sp[4];
  R.lx2 = (s0[0] - sh) * kLog2e;
  R.ly2 = (s0[1] - sh) * kLog2e;
  R.lu2 = (s0[2] - sh) * kLog2e;
  R.lv2 = (s0[3] - sh) * kLog2e;
  R.lw2 = (s4v - sh) * kLog2e;
  R.rng_row = (uint32_t)(bh * L + R.i);
}

// What the mask-only path needs of a query row: the attack query fragment, the normaliser of M, the flags.
template <int DH>
struct MaskRow {
  float qaf[DH / 4];
  float ly2;
  int i;
  bool row_ok, dead;
  uint32_t rng_row;
};
template <int DH>
__device__ __forceinline__ void load_mask_row(const acattn_problem& P, const acattn_bwd_io& IO, const KeyFlags& F, size_t rowbase,
                                              size_t bh, int hoff, int qb, int c, int g, MaskRow<DH>& R) {
  constexpr int KS = DH / 4;
  const int L = P.L, H = P.H;
  R.i = 16 * qb + c;
  R.row_ok = R.i < L;
  const size_t off = (rowbase + (R.row_ok ? R.i : 0)) * H + hoff + KS * g;
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    const f4 ta = *(const f4*)(P.qa + off + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) R.qaf[4 * s4 + e] = R.row_ok ? ta[e] : 0.f;
  }
  R.dead = P.causal ? F.first_valid > R.i : !F.any_valid;
  const float sh = R.dead ? ACATTN_MASK_FILL : 0.f;
  R.ly2 = (IO.row_stats[(bh * L + (R.row_ok ? R.i : 0)) * ACATTN_NSTAT + 1] - sh) * kLog2e;
  R.rng_row = (uint32_t)(bh * L + R.i);
}

__device__ __forceinline__ Consts make_consts(const acattn_problem& P, int DH) {
  Consts K;
  K.sc = P.scalar[0];
  K.s2 = K.sc * K.sc;
  K.inv_sqrt = 1.0f / sqrtf((float)DH);
  K.scale2 = K.inv_sqrt * kLog2e;
  K.nc2 = -(0.5f * K.s2 * K.scale2);
  K.p_drop = P.p_drop;
  K.has_drop = P.p_drop > 0.f;
  K.keep_scale = K.has_drop ? fast_rcp(1.0f - P.p_drop) : 1.0f;
  K.causal = P.causal != 0;
  K.gate_is_prob = P.gate_is_prob;
  K.rkey = rng_key(P.seed + (P.seed_device ? *P.seed_device : 0ull));
  return K;
}

// Key-tile operands in the two shapes the MFMAs want them:
//   row fragment  X[16 t + c][KS g ..]   (A operand of  X . frag^T : scores, dA)
//   col fragment  X[16 t + 4 g + r][16 dt + c]   (A operand of  X^T . tile^T : dq, and with query rows: dk, dv)
template <int DH>
__device__ __forceinline__ void row_frag(const float* X, size_t rowbase, int H, int hoff, int row0, int L, int c, int g,
                                         f4 (&out)[DH / 16]) {
  const float* p = X + (rowbase + min(row0 + c, L - 1)) * H + hoff + (DH / 4) * g;
#pragma unroll
  for (int s4 = 0; s4 < DH / 16; ++s4) out[s4] = *(const f4*)(p + 4 * s4);
}
template <int DH>
__device__ __forceinline__ void col_frag(const float* X, size_t rowbase, int H, int hoff, int row0, int L, int c, int g,
                                         float (&out)[4][DH / 16]) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + 4 * g + r;
    const float* p = X + (rowbase + min(row, L - 1)) * H + hoff + c;
#pragma unroll
    for (int dt = 0; dt < DH / 16; ++dt) out[r][dt] = (X && row < L) ? p[16 * dt] : 0.f;
  }
}
// key halves of the two affines for the lane's 4 keys of a tile, from the K row fragment (lane c holds key 16 t + c)
template <int DH>
__device__ __forceinline__ void key_affine(const f4 (&k4)[DH / 16], const float (&wko)[DH / 4], const float (&wkd)[DH / 4], int g,
                                           f4& co4, f4& cd4) {
  float co = 0.f, cd = 0.f;
#pragma unroll
  for (int s4 = 0; s4 < DH / 16; ++s4)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      co += k4[s4][e] * wko[4 * s4 + e];
      cd += k4[s4][e] * wkd[4 * s4 + e];
    }
  co = quad_sum(co);
  cd = quad_sum(cd);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int src = 4 * (4 * g + r);
    co4[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, co)));
    cd4[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, cd)));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// row kernel
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, bool DCA, bool TWO = false>
__global__ void __launch_bounds__(64, DH >= ACATTN_BWD_ONE_WAVE_DH ? 1 : 2) acattn_bwd_row_kernel(const acattn_problem P, const acattn_bwd_io IO, float* __restrict__ ws) {
  constexpr int KS = DH / 4, DT = DH / 16;
  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;
  const int n_items = P.B * nh;
  const bool causal = P.causal != 0;
  const int rank = blockIdx.x / n_items, item = blockIdx.x - rank * n_items;
  const int qb = causal ? nT - 1 - rank : rank;  // heaviest query blocks first
  int b, h;
  decode_block(item, P.B, nh, b, h);
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * L, bh = (size_t)b * nh + h;
  const int hoff = h * DH;
  const bool full = !IO.attack_only;
  const uint32_t prow_base = (uint32_t)bh * L * (uint32_t)L;

  const KeyFlags F = load_key_flags(P.key_valid, rowbase, L, nT, lane);
  Row<DH> R;
  load_row<DH>(P, IO, F, rowbase, bh, hoff, qb, c, g, R);
  const uint32_t prow = prow_base + (uint32_t)(R.row_ok ? R.i : 0) * (uint32_t)L;
  float* wrow = ws + (bh * L + (R.row_ok ? R.i : 0)) * NSC;

  // a query block none of whose rows carries a cotangent contributes nothing anywhere
  const bool block_active = qblock_active(IO, b, qb);
  if (!block_active) {
    if (R.row_ok) {
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      const uint32_t off = ((uint32_t)rowbase + R.i) * H + hoff + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (full) *(f4*)(IO.dq + off + 16 * dt) = z;
        *(f4*)(IO.dqa + off + 16 * dt) = z;
      }
      if (full && IO.dgate_logits)
        for (int t = 0; t < nT; ++t) store_seg(IO.dgate_logits + prow, 16 * t + 4 * g, L, true, z);
      if (g == 0) {
        *(f4*)wrow = z;
        *(f4*)(wrow + 4) = z;
      }
    }
    return;
  }

  const Consts K = make_consts(P, DH);
  const int i0 = 16 * qb;
  const bool rows_see_a_key = causal ? F.first_valid <= i0 : F.any_valid;
  const int nt = rows_see_a_key ? min(causal ? min(nT, qb + 1) : nT, F.nt_valid) : nT;
#ifndef ACATTN_BWD_NO_MASK_ONLY
  if (!qblock_has_ctx(IO, b, qb)) {
    // ---- [r4] mask-only block (see mask_tile): s M in sweep 1, d Sa -> dqa in sweep 2; dq, the gate rows and the row scalars
    // of the context chain are zero; no parameter partials ---------------------------------------------------------------------
    const float* dmrow_m = IO.d_attack_mask ? IO.d_attack_mask + prow : nullptr;
    const float dpen2_m = IO.d_penalty_part ? 2.0f * IO.d_penalty_part[(size_t)bh * nT + qb] : 0.f;
    // Each tile is one product, four exponentials and a few dozen other instructions: what a wave of this path spends its
    // life on is waiting for the tile's Ka rows.  So the tiles are unrolled (13 = ceil(208 / 16)), the rows of tile t + 1
    // are requested before tile t is worked on (two register sets, roles fixed per tile), and Mt / d M of every tile stay in
    // registers (104 of them: this path has the room), so sweep 2 is the dqa product alone, its Ka columns prefetched alike.
    constexpr int NTM = 13;
    f4 Mt[NTM], dMk[NTM];
    f4 ka_a[DT], ka_b[DT], dme_a = {0.f, 0.f, 0.f, 0.f}, dme_b = dme_a;
    row_frag<DH>(P.ka, rowbase, H, hoff, 0, L, c, g, ka_a);
    if (dmrow_m) dme_a = load_seg(dmrow_m, 4 * g, L, R.row_ok);
    float s_m = 0.f;
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
      if (t < nt) {
        f4(&cur)[DT] = (t & 1) ? ka_b : ka_a;
        f4(&nxt)[DT] = (t & 1) ? ka_a : ka_b;
        const f4 dcur = (t & 1) ? dme_b : dme_a;
        if (t + 1 < nt) {
          row_frag<DH>(P.ka, rowbase, H, hoff, 16 * (t + 1), L, c, g, nxt);
          if (dmrow_m) ((t & 1) ? dme_a : dme_b) = load_seg(dmrow_m, 16 * (t + 1) + 4 * g, L, R.row_ok);
        }
        f4 aM = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < KS / 4; ++s4)
#pragma unroll
          for (int e = 0; e < 4; ++e) aM = mfma16(cur[s4][e], R.qaf[4 * s4 + e], aM);
        uint32_t eb4, ab4;
        tile_bits(F, t, g, R.i, L, causal, R.row_ok, R.dead, eb4, ab4);
        mask_tile(R, K, aM, t, g, eb4, dcur, dpen2_m, Mt[t], dMk[t]);
        s_m += hsum(Mt[t] * dMk[t]);
      }
    }
    const float sM = quad_sum(s_m);
    if (R.row_ok && g == 0) {
      *(f4*)wrow = f4{0.f, 0.f, 0.f, 0.f};
      *(f4*)(wrow + 4) = f4{sM, 0.f, 0.f, 0.f};
    }
    f4 oqa[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oqa[dt] = f4{0.f, 0.f, 0.f, 0.f};
    float kc_a[4][DT], kc_b[4][DT];
    col_frag<DH>(P.ka, rowbase, H, hoff, 0, L, c, g, kc_a);
#pragma unroll
    for (int t = 0; t < NTM; ++t) {
      if (t < nt) {
        float(&cur)[4][DT] = (t & 1) ? kc_b : kc_a;
        float(&nxt)[4][DT] = (t & 1) ? kc_a : kc_b;
        if (t + 1 < nt) col_frag<DH>(P.ka, rowbase, H, hoff, 16 * (t + 1), L, c, g, nxt);
        const f4 dSa = (Mt[t] * (dMk[t] - sM)) * K.inv_sqrt;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) oqa[dt] = mfma16(cur[r][dt], dSa[r], oqa[dt]);
      }
    }
    if (R.row_ok) {
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      const uint32_t off = ((uint32_t)rowbase + R.i) * H + hoff + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (full) *(f4*)(IO.dq + off + 16 * dt) = z;
        *(f4*)(IO.dqa + off + 16 * dt) = oqa[dt];
      }
      if (full && IO.dgate_logits && !IO.dgate_summed)
        for (int t = 0; t < nT; ++t) store_seg(IO.dgate_logits + prow, 16 * t + 4 * g, L, true, z);
    }
    return;
  }
#endif
  float wko[KS], wkd[KS];
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    const f4 ak = *(const f4*)(P.w_order + DH + KS * g + 4 * s4), dk = *(const f4*)(P.w_dist + DH + KS * g + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wko[4 * s4 + e] = ak[e] * -kLog2e;
      wkd[4 * s4 + e] = dk[e];
    }
  }
  const bool order_select = !causal || __ballot(R.dead) != 0ull;
  const float* grow = P.gate_logits + (rowbase + (R.row_ok ? R.i : 0)) * L;
  const float* dmrow = IO.d_attack_mask ? IO.d_attack_mask + prow : nullptr;
  // the mask penalty's cotangent formed from the rebuilt tile (acattn_bwd_io.d_penalty_part): d M += 2 d_pen (M - 1)
  const float dpen2 = IO.d_penalty_part ? 2.0f * IO.d_penalty_part[(size_t)bh * nT + qb] : 0.f;

  auto build = [&](int t, Tile& T) {
    f4 k4[DT], ka4[DT], v4[DT];
    row_frag<DH>(P.k, rowbase, H, hoff, 16 * t, L, c, g, k4);
    row_frag<DH>(P.ka, rowbase, H, hoff, 16 * t, L, c, g, ka4);
    row_frag<DH>(P.v, rowbase, H, hoff, 16 * t, L, c, g, v4);
    const f4 gl = load_seg(grow, 16 * t + 4 * g, L, R.row_ok);
    f4 co4, cd4;
    key_affine<DH>(k4, wko, wkd, g, co4, cd4);
    uint32_t eb4, ab4;
    tile_bits(F, t, g, R.i, L, causal, R.row_ok, R.dead, eb4, ab4);
    tile_forward<DH, DCA, TWO>(R, K, k4, ka4, v4, co4, cd4, gl, t, g, eb4, ab4, order_select, T);
  };
  const float dpen2_2 = (TWO && IO.d_penalty_part2) ? 2.0f * IO.d_penalty_part2[(size_t)bh * nT + qb] : 0.f;  // second set
  float s_dc2 = 0.f, s_r1a2 = 0.f, cM0_2 = 0.f;

  // ---- sweep 1: the twelve row sums ---------------------------------------------------------------------------------
  float s_da = 0.f, s_dc = 0.f, s_r1a = 0.f, s_r1b = 0.f, cP0 = 0.f, cPc = 0.f, cPr = 0.f, cPa = 0.f, cM0 = 0.f, cMc = 0.f,
        cMr = 0.f, cMa = 0.f;
  for (int t = 0; t < nt; ++t) {
    Tile T;
    build(t, T);
    const f4 dMout = (dmrow ? load_seg(dmrow, 16 * t + 4 * g, L, R.row_ok) : f4{0.f, 0.f, 0.f, 0.f}) + (T.M - 1.0f) * dpen2;
    const f4 apd = T.Ap * T.dAp;
    const f4 alpha = T.Aw * T.dAw, beta = T.Aw;
    const f4 gam = T.Ac * (1.0f - T.gt);
    const f4 PE = T.P * T.ex1;
    s_dc += hsum(alpha);
    s_r1a += hsum(gam * alpha);
    s_r1b += hsum(gam * beta);
    cPc += hsum(T.P * (T.gt * beta + T.ex1 * (gam * beta)));
    cPr += hsum(PE * T.Ac);
    cMc += hsum(T.M * (PE * (gam * beta)));
    cMr += hsum(T.M * (PE * T.Ac));
    if constexpr (DCA) {
      s_da += hsum(apd);
      cP0 += hsum(T.P * (T.gt * alpha + T.ex1 * (gam * alpha) + T.M * apd));
      cPa += hsum(T.P * (T.M * T.Ap));
      cM0 += hsum(T.M * ((T.P - T.nz) * apd - PE * (gam * alpha) + dMout));
      cMa += hsum(T.M * ((T.P - T.nz) * T.Ap));
    } else {  // d A_p = 0: da = 0 and every term it multiplies drops out
      cP0 += hsum(T.P * (T.gt * alpha + T.ex1 * (gam * alpha)));
      cM0 += hsum(T.M * (dMout - PE * (gam * alpha)));
    }
    if constexpr (TWO) {  // the second set's three sums (its cotangent-free coefficients are the first set's)
      const f4 alpha2 = T.Aw * T.dAw2;
      s_dc2 += hsum(alpha2);
      s_r1a2 += hsum(gam * alpha2);
      cM0_2 += hsum(T.M * ((T.M - 1.0f) * dpen2_2 - PE * (gam * alpha2)));
    }
  }
  const float da = quad_sum(s_da), dc = quad_sum(s_dc);
  const float r1 = quad_sum(s_r1a) - dc * quad_sum(s_r1b);  // (quad_sum of the same lane sums again below: registers, not time)
  const float sP = quad_sum(cP0) - dc * quad_sum(cPc) - r1 * quad_sum(cPr) - da * quad_sum(cPa);
  const float q_r1b = quad_sum(s_r1b), q_cMc = quad_sum(cMc), q_cMr = quad_sum(cMr);
  const float sM = quad_sum(cM0) + dc * q_cMc + r1 * q_cMr - da * quad_sum(cMa);
  float dc2 = 0.f, r1_2 = 0.f, sM2 = 0.f;
  if constexpr (TWO) {
    dc2 = quad_sum(s_dc2);
    r1_2 = quad_sum(s_r1a2) - dc2 * q_r1b;
    sM2 = quad_sum(cM0_2) + dc2 * q_cMc + r1_2 * q_cMr;
  }
  if (R.row_ok && g == 0) {
    *(f4*)wrow = f4{da, dc, r1, sP};
    *(f4*)(wrow + 4) = f4{sM, dc2, r1_2, sM2};
  }

  // ---- sweep 2: dS, dSa -> dq, dqa; gate partials; query halves of the parameter gradients -------------------------------
  f4 oq[DT], oqa[DT], oqa2[TWO ? DT : 1];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    oq[dt] = f4{0.f, 0.f, 0.f, 0.f};
    oqa[dt] = oq[dt];
    if constexpr (TWO) oqa2[dt] = oq[dt];
  }
  float da_o = 0.f, da_d = 0.f, dsc_acc = 0.f;
  for (int t = 0; t < nt; ++t) {
    Tile T;
    build(t, T);
    const int j0 = 16 * t + 4 * g;
    const f4 dMout = (dmrow ? load_seg(dmrow, j0, L, R.row_ok) : f4{0.f, 0.f, 0.f, 0.f}) + (T.M - 1.0f) * dpen2;
    f4 dS, dSa, dgl, d_o, d_d;
    float dsc;
    tile_backward<DCA>(T, K, da, dc, r1, sP, sM, dMout, R.i, j0, dS, dSa, dgl, d_o, d_d, dsc);
    if (full && IO.dgate_logits) store_seg(IO.dgate_logits + prow, j0, L, R.row_ok, dgl);
    da_o += hsum(d_o);
    da_d += hsum(d_d);
    dsc_acc += dsc;
    float kc[4][DT], kac[4][DT];
    if (full) col_frag<DH>(P.k, rowbase, H, hoff, 16 * t, L, c, g, kc);
    col_frag<DH>(P.ka, rowbase, H, hoff, 16 * t, L, c, g, kac);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (full) oq[dt] = mfma16(kc[r][dt], dS[r], oq[dt]);
        oqa[dt] = mfma16(kac[r][dt], dSa[r], oqa[dt]);
      }
    if constexpr (TWO) {
      const f4 dSa2 = second_set_dsa(T, K, dc2, r1_2, sM2, (T.M - 1.0f) * dpen2_2);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) oqa2[dt] = mfma16(kac[r][dt], dSa2[r], oqa2[dt]);
    }
  }
  if (full && IO.dgate_logits)
    for (int t = nt; t < nT; ++t) store_seg(IO.dgate_logits + prow, 16 * t + 4 * g, L, R.row_ok, f4{0.f, 0.f, 0.f, 0.f});
  da_o = quad_sum(da_o);
  da_d = quad_sum(da_d);
  dsc_acc = quad_sum(dsc_acc);
  if (R.row_ok) {
    const uint32_t off = ((uint32_t)rowbase + R.i) * H + hoff + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      // rank-1 terms of dq: query halves of the affine weights in the lane's output-column order
      const f4 wo_lo = *(const f4*)(P.w_order + 16 * dt + 4 * g), wd_lo = *(const f4*)(P.w_dist + 16 * dt + 4 * g);
      if (full) *(f4*)(IO.dq + off + 16 * dt) = oq[dt] + da_o * wo_lo + da_d * wd_lo;
      *(f4*)(IO.dqa + off + 16 * dt) = oqa[dt];
      if constexpr (TWO) *(f4*)(IO.dqa2 + off + 16 * dt) = oqa2[dt];
    }
  }
  if (!full) return;  // the parameter partials are not read either
  // query halves of dw_order / dw_dist, db_order, db_dist, d scalar: summed over the block's 16 rows, then added to the
  // (sequence, head) partial row (a few hundred float atomics per launch row; the rows were zeroed by the launcher)
  const int stride_w = IO.part_stride ? IO.part_stride : 2 * DH, stride_s = IO.part_stride ? IO.part_stride : 4;
  const float ro = R.row_ok ? 1.0f : 0.0f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float vo = row16_sum(da_o * R.qf[s] * ro), vd = row16_sum(da_d * R.qf[s] * ro);
    if (c == 0) {
      atomicAdd(IO.dw_order_part + bh * stride_w + KS * g + s, vo);
      atomicAdd(IO.dw_dist_part + bh * stride_w + KS * g + s, vd);
    }
  }
  const float so = row16_sum(da_o * ro), sd = row16_sum(da_d * ro), ss = row16_sum(dsc_acc * ro);
  if (lane == 0) {
    atomicAdd(IO.dsmall_part + bh * stride_s + 0, so);
    atomicAdd(IO.dsmall_part + bh * stride_s + 1, sd);
    atomicAdd(IO.dsmall_part + bh * stride_s + 2, ss);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// key kernel
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, bool DCA, bool TWO = false>
__global__ void __launch_bounds__(64, DH >= ACATTN_BWD_ONE_WAVE_DH ? 1 : 2) acattn_bwd_key_kernel(const acattn_problem P, const acattn_bwd_io IO, const float* __restrict__ ws) {
  constexpr int KS = DH / 4, DT = DH / 16;
  constexpr int TS = 20;  // row stride of a transposed 16 x 16 tile in LDS (16-byte aligned rows, conflict-free reads)
  __shared__ __attribute__((aligned(16))) float tr[4][16 * TS + 32];
  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;
  const int n_items = P.B * nh;
  const bool causal = P.causal != 0;
  const int rank = blockIdx.x / n_items, item = blockIdx.x - rank * n_items;
  const int t = rank;  // under the causal mask key tile 0 is seen by every query block: heaviest first
  int b, h;
  decode_block(item, P.B, nh, b, h);
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * L, bh = (size_t)b * nh + h;
  const int hoff = h * DH;
  const bool full = !IO.attack_only;
  const uint32_t prow_base = (uint32_t)bh * L * (uint32_t)L;

  const KeyFlags F = load_key_flags(P.key_valid, rowbase, L, nT, lane);
  const Consts K = make_consts(P, DH);
  // this wave's 16 keys: row fragments of K, Ka, V (fixed), key halves of the affines
  f4 k4[DT], ka4[DT], v4[DT];
  row_frag<DH>(P.k, rowbase, H, hoff, 16 * t, L, c, g, k4);
  row_frag<DH>(P.ka, rowbase, H, hoff, 16 * t, L, c, g, ka4);
  row_frag<DH>(P.v, rowbase, H, hoff, 16 * t, L, c, g, v4);
  float wko[KS], wkd[KS];
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    const f4 ak = *(const f4*)(P.w_order + DH + KS * g + 4 * s4), dk = *(const f4*)(P.w_dist + DH + KS * g + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      wko[4 * s4 + e] = ak[e] * -kLog2e;
      wkd[4 * s4 + e] = dk[e];
    }
  }
  f4 co4, cd4;
  key_affine<DH>(k4, wko, wkd, g, co4, cd4);

  f4 aK[DT], aKa[DT], aV[DT];  // dK^T, dKa^T, dV^T: lane (c, g) holds key 16 t + c, columns 16 dt + 4 g ..
  f4 aKa2[TWO ? DT : 1];       // [r4] dKa^T of the second cotangent set
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    aK[dt] = f4{0.f, 0.f, 0.f, 0.f};
    aKa[dt] = aK[dt];
    aV[dt] = aK[dt];
    if constexpr (TWO) aKa2[dt] = aK[dt];
  }
  f4 dco = {0.f, 0.f, 0.f, 0.f}, dcd = dco;  // d (key half of the affines) of keys 16 t + 4 g + r, summed over queries at the end

  // query blocks that see this key tile: under the causal mask qb >= t; a tile past the last real item is only seen
  // by blocks with a dead row (they spread over every key)
  const int qb_lo = causal ? t : 0;
  for (int qb = qb_lo; qb < nT; ++qb) {
    const int i0 = 16 * qb;
    const bool rows_see_a_key = causal ? F.first_valid <= i0 : F.any_valid;
    const int nt_q = rows_see_a_key ? min(causal ? min(nT, qb + 1) : nT, F.nt_valid) : nT;
    if (t >= nt_q) continue;  // the row kernel skipped this pair too: it carries no probability mass
    if (!qblock_active(IO, b, qb)) continue;
#ifndef ACATTN_BWD_NO_MASK_ONLY
    if (!qblock_has_ctx(IO, b, qb)) {  // [r4] mask-only pair (mask_tile): d Sa alone, into dKa
      MaskRow<DH> R;
      load_mask_row<DH>(P, IO, F, rowbase, bh, hoff, qb, c, g, R);
      float qac[4][DT];  // every load of the pair is requested here, in front of the first wait: one round trip per pair
      col_frag<DH>(P.qa, rowbase, H, hoff, i0, L, c, g, qac);
      const float sM = ws[(bh * L + (R.row_ok ? R.i : 0)) * NSC + 4];
      const uint32_t prow = prow_base + (uint32_t)(R.row_ok ? R.i : 0) * (uint32_t)L;
      const int j0 = 16 * t + 4 * g;
      f4 aM = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < KS / 4; ++s4)
#pragma unroll
        for (int e = 0; e < 4; ++e) aM = mfma16(ka4[s4][e], R.qaf[4 * s4 + e], aM);
      uint32_t eb4, ab4;
      tile_bits(F, t, g, R.i, L, causal, R.row_ok, R.dead, eb4, ab4);
      const f4 dMext = IO.d_attack_mask ? load_seg(IO.d_attack_mask + prow, j0, L, R.row_ok) : f4{0.f, 0.f, 0.f, 0.f};
      const float dpen2 = IO.d_penalty_part ? 2.0f * IO.d_penalty_part[(size_t)bh * nT + qb] : 0.f;
      f4 Mt, dM;
      mask_tile(R, K, aM, t, g, eb4, dMext, dpen2, Mt, dM);
      const f4 dSa = (Mt * (dM - sM)) * K.inv_sqrt;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      *(f4*)(&tr[1][c * TS + 4 * g]) = dSa;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float b_sa = tr[1][(4 * g + s) * TS + c];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) aKa[dt] = mfma16(qac[s][dt], b_sa, aKa[dt]);
      }
      continue;
    }
#endif
    Row<DH> R;
    load_row<DH>(P, IO, F, rowbase, bh, hoff, qb, c, g, R);
    const bool order_select = !causal || __ballot(R.dead) != 0ull;
    const float* wrow = ws + (bh * L + (R.row_ok ? R.i : 0)) * NSC;
    const f4 w0 = *(const f4*)wrow;
    const float sM = wrow[4];
    const uint32_t prow = prow_base + (uint32_t)(R.row_ok ? R.i : 0) * (uint32_t)L;
    const int j0 = 16 * t + 4 * g;
    const f4 gl = load_seg(P.gate_logits + (rowbase + (R.row_ok ? R.i : 0)) * L, j0, L, R.row_ok);
    f4 dMout = IO.d_attack_mask ? load_seg(IO.d_attack_mask + prow, j0, L, R.row_ok) : f4{0.f, 0.f, 0.f, 0.f};
    uint32_t eb4, ab4;
    tile_bits(F, t, g, R.i, L, causal, R.row_ok, R.dead, eb4, ab4);
    Tile T;
    tile_forward<DH, DCA, TWO>(R, K, k4, ka4, v4, co4, cd4, gl, t, g, eb4, ab4, order_select, T);
    if (IO.d_penalty_part) dMout += (T.M - 1.0f) * (2.0f * IO.d_penalty_part[(size_t)bh * nT + qb]);
    f4 dS, dSa, dgl, d_o, d_d;
    float dsc;
    tile_backward<DCA>(T, K, w0[0], w0[1], w0[2], w0[3], sM, dMout, R.i, j0, dS, dSa, dgl, d_o, d_d, dsc);
    dco += d_o;
    dcd += d_d;
    // turn the four [query c][key 4 g + r] tiles so that the query index becomes the MFMA reduction index
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    *(f4*)(&tr[0][c * TS + 4 * g]) = dS;
    *(f4*)(&tr[1][c * TS + 4 * g]) = dSa;
    if constexpr (DCA) *(f4*)(&tr[2][c * TS + 4 * g]) = T.Ap;
    if constexpr (TWO && !DCA) {  // the second set's d Sa takes the slot A_p does not need (TWO comes without d ctx_attacked)
      const float dpen2_2 = IO.d_penalty_part2 ? 2.0f * IO.d_penalty_part2[(size_t)bh * nT + qb] : 0.f;
      *(f4*)(&tr[2][c * TS + 4 * g]) = second_set_dsa(T, K, wrow[5], wrow[6], wrow[7], (T.M - 1.0f) * dpen2_2);
    }
    *(f4*)(&tr[3][c * TS + 4 * g]) = T.Aw;
    float qc[4][DT], qac[4][DT], gac[4][DT], gcc[4][DT];
    if (full) col_frag<DH>(P.q, rowbase, H, hoff, i0, L, c, g, qc);
    col_frag<DH>(P.qa, rowbase, H, hoff, i0, L, c, g, qac);
    if constexpr (DCA)
      if (full) col_frag<DH>(IO.d_ctx_attacked, rowbase, H, hoff, i0, L, c, g, gac);
    if (full) col_frag<DH>(IO.d_ctx_calibrated, rowbase, H, hoff, i0, L, c, g, gcc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float b_s = tr[0][(4 * g + s) * TS + c], b_sa = tr[1][(4 * g + s) * TS + c];
      const float b_aw = tr[3][(4 * g + s) * TS + c];
      float b_ap = 0.f;
      if constexpr (DCA) b_ap = tr[2][(4 * g + s) * TS + c];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (full) {
          aK[dt] = mfma16(qc[s][dt], b_s, aK[dt]);
          if constexpr (DCA) aV[dt] = mfma16(gac[s][dt], b_ap, aV[dt]);
          aV[dt] = mfma16(gcc[s][dt], b_aw, aV[dt]);
        }
        aKa[dt] = mfma16(qac[s][dt], b_sa, aKa[dt]);
        if constexpr (TWO && !DCA) aKa2[dt] = mfma16(qac[s][dt], tr[2][(4 * g + s) * TS + c], aKa2[dt]);
      }
    }
  }

  // ---- results: dk (+ rank-1 key-half terms), dka, dv; key halves of the parameter gradients --------------------------------
  // d co_j, d cd_j: sum over the query lanes; then every lane needs the value of ITS output key 16 t + c
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float so = row16_sum(dco[r]), sd = row16_sum(dcd[r]);
    if (c == 0) {
      tr[0][4 * g + r] = so;
      tr[1][4 * g + r] = sd;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float dco_c = tr[0][c], dcd_c = tr[1][c];
  const int key = 16 * t + c;
  if (key < L) {
    const size_t o = (rowbase + key) * H + hoff + 4 * g;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      *(f4*)(IO.dka + o + 16 * dt) = aKa[dt];
      if constexpr (TWO) *(f4*)(IO.dka2 + o + 16 * dt) = aKa2[dt];
      if (full) {
        const f4 wo_hi = *(const f4*)(P.w_order + DH + 16 * dt + 4 * g), wd_hi = *(const f4*)(P.w_dist + DH + 16 * dt + 4 * g);
        *(f4*)(IO.dk + o + 16 * dt) = aK[dt] + dco_c * wo_hi + dcd_c * wd_hi;
        *(f4*)(IO.dv + o + 16 * dt) = aV[dt];
      }
    }
  }
  if (!full) return;
  // dw_order[dh:] += sum_j d co_j K_j (natural units: co was pre-scaled by -log2e inside the sigmoid argument only)
  const int stride_w = IO.part_stride ? IO.part_stride : 2 * DH;
  const float okk = key < L ? 1.0f : 0.0f;
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float vo = row16_sum(dco_c * k4[s4][e] * okk), vd = row16_sum(dcd_c * k4[s4][e] * okk);
      if (c == 0) {
        atomicAdd(IO.dw_order_part + bh * stride_w + DH + KS * g + 4 * s4 + e, vo);
        atomicAdd(IO.dw_dist_part + bh * stride_w + DH + KS * g + 4 * s4 + e, vd);
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// one-row kernel
// ---------------------------------------------------------------------------------------------------------------------
// The last layer is read at ONE position per sequence (item_seq_len - 1, abstract_recommender.py:130-134), so in the
// calibrated-loss pass its context cotangents are zero everywhere but in that row: of the L x L backward only one row
// of every probability tensor matters.  One wave per (sequence, head): lane l owns keys 4 l .. 4 l + 3 of that row
// (the lane's 4 keys of tile_elementwise / tile_backward, with t = l / 4, g = l % 4 -- the same code, the same random
// draws), the four products of a key are plain dot products against row vectors kept in LDS, the row sums are wave
// reductions.  Every output is written in full (rows other than the read one are zero), so the caller needs no fill.
struct RowScalars {
  float ao2, ad, lx2, ly2, lu2, lv2, lw2;
  int i;
  uint32_t rng_row;
};

__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// ACC: dqa (all rows) and dka already hold the contribution of the mask cotangent (the mask-only path of the
// row-resident kernel, which is linear in d M and independent of the context cotangents): the read row's chain is
// added to them instead of overwriting.
#ifdef ACATTN_ONEROW_STAMPS
__device__ unsigned long long g_onerow_stamps[2048 * 8];
#define OR_STAMP(k)                                                                \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    unsigned long long now_;                                                       \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); \
    if ((k) >= 0) cyc_[(k) >= 0 ? (k) : 0] += now_ - last_;                        \
    last_ = now_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define OR_STAMP(k)
#endif

template <int DH, bool ACC, bool SLICED>
__global__ void __launch_bounds__(64) acattn_bwd_onerow_kernel(const acattn_problem P, const acattn_bwd_io IO) {
#ifdef ACATTN_ONEROW_STAMPS
  unsigned long long cyc_[8] = {}, last_ = 0;
#endif
  OR_STAMP(-1);
  constexpr int D4 = DH / 4;
  __shared__ __attribute__((aligned(16))) float vec[8][DH];  // q, qa, d_ctx_att, d_ctx_cal, wko, wkd, w_order_q, w_dist_q
  __shared__ __attribute__((aligned(16))) float col[4][256];  // dS, dSa, d_o, d_d of every key
  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;
  const bool causal = P.causal != 0;
  int b, h;
  decode_block(blockIdx.x, P.B, nh, b, h);
  const int lane = threadIdx.x;
  const size_t rowbase = (size_t)b * L, bh = (size_t)b * nh + h;
  const int hoff = h * DH;
  // n_read_rows == 1.  Positions outside [0, L) (item_seq_len == 0 gives -1) are a caller error the reference's gather
  // answers with an index error; here they are clamped, so the launch at least stays inside the sequence's memory
  const int i = min(max((int)IO.read_rows[b], 0), L - 1);
  const KeyFlags F = load_key_flags(P.key_valid, rowbase, L, nT, lane);
  const Consts K = make_consts(P, DH);

  // ---- the row's vectors ----------------------------------------------------------------------------------------------
  for (int d = lane; d < DH; d += 64) {
    const size_t o = (rowbase + i) * H + hoff + d;
    vec[0][d] = P.q[o];
    vec[1][d] = P.qa[o];
    vec[2][d] = IO.d_ctx_attacked ? IO.d_ctx_attacked[o] : 0.f;
    vec[3][d] = IO.d_ctx_calibrated ? IO.d_ctx_calibrated[o] : 0.f;
    vec[4][d] = P.w_order[DH + d] * -kLog2e;
    vec[5][d] = P.w_dist[DH + d];
    vec[6][d] = P.w_order[d];
    vec[7][d] = P.w_dist[d];
  }
  __syncthreads();
  OR_STAMP(0);
  RowScalars R;
  {
    float ao = 0.f, adv = 0.f;
#pragma unroll
    for (int d = 0; d < DH; ++d) {
      ao += vec[0][d] * vec[6][d];
      adv += vec[0][d] * vec[7][d];
    }
    R.ao2 = -kLog2e * (ao + P.b_order[0]);
    R.ad = adv + P.b_dist[0];
  }
  const bool dead = causal ? F.first_valid > i : !F.any_valid;
  {
    const float sh = dead ? ACATTN_MASK_FILL : 0.f;
    const float* sp = IO.row_stats + (bh * L + i) * ACATTN_NSTAT;
    R.lx2 = (sp[0] - sh) * kLog2e;
    R.ly2 = (sp[1] - sh) * kLog2e;
    R.lu2 = (sp[2] - sh) * kLog2e;
    R.lv2 = (sp[3] - sh) * kLog2e;
    R.lw2 = (sp[4] - sh) * kLog2e;
  }
  R.i = i;
  R.rng_row = (uint32_t)(bh * L + i);
  const bool order_select = !causal || dead;

  // ---- the lane's 4 keys: S, Sa, dA_p, dA_w and the key halves of the affines as dot products ---------------------------
  // L <= 64: 16 key groups x 4 slices of the head dimension (lane = 16 * slice + group: the four lanes of a group each
  // read a quarter of every row and meet in a 4-lane sum; all of them then carry the group's values, the slice-0 lanes
  // alone feed the wave sums).  Longer rows: one key group per lane, the whole head dimension.
  constexpr bool sliced = SLICED;  // L <= 64
  constexpr int N4 = SLICED ? D4 / 4 : D4;            // float4 columns of a row this lane works on
  const int kgrp = sliced ? (lane & 15) : lane;       // key group: keys 4 kgrp .. 4 kgrp + 3
  const int slice = sliced ? (lane >> 4) : 0;
  const int d_lo = slice * N4;
  const float lead = slice == 0 ? 1.0f : 0.0f;
  const int t = kgrp >> 2, g = kgrp & 3, j0 = 4 * kgrp;
  f4 aS = {0.f, 0.f, 0.f, 0.f}, aM = aS, aP = aS, aW = aS, co4 = aS, cd4 = aS;
  if constexpr (SLICED) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = (rowbase + min(j0 + r, L - 1)) * H + hoff;
#pragma unroll
      for (int k4 = 0; k4 < N4; ++k4) {  // (compile-time trip count: every load of the lane is in flight at once)
        const int d4 = d_lo + k4;
        const f4 kv = *(const f4*)(P.k + o + 4 * d4), kav = *(const f4*)(P.ka + o + 4 * d4), vv = *(const f4*)(P.v + o + 4 * d4);
        const f4 q4 = *(const f4*)(&vec[0][4 * d4]), qa4 = *(const f4*)(&vec[1][4 * d4]);
        const f4 ga4 = *(const f4*)(&vec[2][4 * d4]), gc4 = *(const f4*)(&vec[3][4 * d4]);
        const f4 wo4 = *(const f4*)(&vec[4][4 * d4]), wd4 = *(const f4*)(&vec[5][4 * d4]);
        aS[r] += hsum(kv * q4);
        aM[r] += hsum(kav * qa4);
        aP[r] += hsum(vv * ga4);
        aW[r] += hsum(vv * gc4);
        co4[r] += hsum(kv * wo4);
        cd4[r] += hsum(kv * wd4);
      }
    }
  } else {
    // [r4] long rows (one key group per lane, the whole head dimension): column piece by column piece -- the row's six
    // vectors are read from LDS once per piece, the 12 row pieces of the lane's 4 keys are in flight together, and the fence
    // keeps the scheduler from hoisting every piece's loads to the top (that form needed 512 registers + 277 spilled and
    // 1.2 KB of scratch per lane at head size 32: 166 us per launch at B = 512, L = 200)
    size_t o4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o4[r] = (rowbase + min(j0 + r, L - 1)) * H + hoff;
#pragma unroll 2
    for (int k4 = 0; k4 < N4; ++k4) {  // a real loop: two pieces (24 row loads) in flight per trip
      const f4 q4 = *(const f4*)(&vec[0][4 * k4]), qa4 = *(const f4*)(&vec[1][4 * k4]);
      const f4 ga4 = *(const f4*)(&vec[2][4 * k4]), gc4 = *(const f4*)(&vec[3][4 * k4]);
      const f4 wo4 = *(const f4*)(&vec[4][4 * k4]), wd4 = *(const f4*)(&vec[5][4 * k4]);
      f4 kv[4], kav[4], vv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        kv[r] = *(const f4*)(P.k + o4[r] + 4 * k4);
        kav[r] = *(const f4*)(P.ka + o4[r] + 4 * k4);
        vv[r] = *(const f4*)(P.v + o4[r] + 4 * k4);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        aS[r] += hsum(kv[r] * q4);
        aM[r] += hsum(kav[r] * qa4);
        aP[r] += hsum(vv[r] * ga4);
        aW[r] += hsum(vv[r] * gc4);
        co4[r] += hsum(kv[r] * wo4);
        cd4[r] += hsum(kv[r] * wd4);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (sliced) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      aS[r] = quad_sum(aS[r]);
      aM[r] = quad_sum(aM[r]);
      aP[r] = quad_sum(aP[r]);
      aW[r] = quad_sum(aW[r]);
      co4[r] = quad_sum(co4[r]);
      cd4[r] = quad_sum(cd4[r]);
    }
  }
  OR_STAMP(1);
  const f4 gl = load_seg(P.gate_logits + (rowbase + i) * L, j0, L, true);
  uint32_t eb4, ab4;
  tile_bits(F, t, g, i, L, causal, true, dead, eb4, ab4);
  if (j0 >= 16 * nT) eb4 = 0u;  // lanes past the last key tile
  Tile T;
  tile_elementwise(R, K, aS, aM, aP, aW, co4, cd4, gl, t, g, eb4, ab4, order_select, T);

  // ---- row scalars of the chained soft-max backward (section 4.2 of DESIGN.md, directly: the row is in registers) ----------
  const float da = wave_sum(lead * hsum(T.Ap * T.dAp)), dc = wave_sum(lead * hsum(T.Aw * T.dAw));
  const f4 dw = T.Aw * (T.dAw - dc);
  const f4 dac = (1.0f - T.gt) * dw;
  const float r1 = wave_sum(lead * hsum(T.Ac * dac));
  const f4 du = T.Ap * (T.dAp - da), dv = T.Ac * (dac - r1);
  f4 dP = T.gt * dw + dv * T.ex1 + du * T.M;
  f4 dM = du * (T.P - T.nz) - dv * (T.P * T.ex1);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    dP[r] = keep_and(dP[r] * K.keep_scale, T.ka, r);
    dM[r] = keep_and(dM[r] * K.keep_scale, T.km, r);
  }
  const float sP = wave_sum(lead * hsum(T.Pt * dP)), sM = wave_sum(lead * hsum(T.Mt * dM));
  f4 dS, dSa, dgl, d_o, d_d;
  float dsc;
  tile_backward(T, K, da, dc, r1, sP, sM, f4{0.f, 0.f, 0.f, 0.f}, i, j0, dS, dSa, dgl, d_o, d_d, dsc);
  const float da_o = wave_sum(lead * hsum(d_o)), da_d = wave_sum(lead * hsum(d_d)), dsc_sum = wave_sum(lead * dsc);
  if (slice == 0) {
    *(f4*)(&col[0][j0]) = dS;
    *(f4*)(&col[1][j0]) = dSa;
    *(f4*)(&col[2][j0]) = d_o;
    *(f4*)(&col[3][j0]) = d_d;
  }

  OR_STAMP(2);
  // ---- key side: dk, dka, dv of the lane's 4 keys (rank one in the row's vectors) ------------------------------------------
  if constexpr (SLICED) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = j0 + r;
      if (j < L) {
        const size_t o = (rowbase + j) * H + hoff;
#pragma unroll
        for (int k4 = 0; k4 < N4; ++k4) {
          const int d4 = d_lo + k4;
          const f4 q4 = *(const f4*)(&vec[0][4 * d4]), qa4 = *(const f4*)(&vec[1][4 * d4]);
          const f4 ga4 = *(const f4*)(&vec[2][4 * d4]), gc4 = *(const f4*)(&vec[3][4 * d4]);
          const f4 wo_hi = *(const f4*)(P.w_order + DH + 4 * d4), wd_hi = *(const f4*)(P.w_dist + DH + 4 * d4);
          *(f4*)(IO.dk + o + 4 * d4) = q4 * dS[r] + wo_hi * d_o[r] + wd_hi * d_d[r];
          if (ACC)
            *(f4*)(IO.dka + o + 4 * d4) += qa4 * dSa[r];
          else
            *(f4*)(IO.dka + o + 4 * d4) = qa4 * dSa[r];
          *(f4*)(IO.dv + o + 4 * d4) = ga4 * T.Ap[r] + gc4 * T.Aw[r];
        }
      }
    }
  } else {  // [r4] column piece by column piece, as above
#pragma unroll 2
    for (int k4 = 0; k4 < N4; ++k4) {
      const f4 q4 = *(const f4*)(&vec[0][4 * k4]), qa4 = *(const f4*)(&vec[1][4 * k4]);
      const f4 ga4 = *(const f4*)(&vec[2][4 * k4]), gc4 = *(const f4*)(&vec[3][4 * k4]);
      const f4 wo_hi = *(const f4*)(P.w_order + DH + 4 * k4), wd_hi = *(const f4*)(P.w_dist + DH + 4 * k4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = j0 + r;
        if (j < L) {
          const size_t o = (rowbase + j) * H + hoff;
          *(f4*)(IO.dk + o + 4 * k4) = q4 * dS[r] + wo_hi * d_o[r] + wd_hi * d_d[r];
          if (ACC)
            *(f4*)(IO.dka + o + 4 * k4) += qa4 * dSa[r];
          else
            *(f4*)(IO.dka + o + 4 * k4) = qa4 * dSa[r];
          *(f4*)(IO.dv + o + 4 * k4) = ga4 * T.Ap[r] + gc4 * T.Aw[r];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  OR_STAMP(3);
  // ---- gate-logit gradient: the read row carries dgl, every other row is zero --------------------------------------------------
  if (IO.dgate_logits && IO.dgate_summed) {
    // [r4] head-summed form ([B,L,L], zeroed by the launcher): this head's read row is ADDED, nothing else is touched
    if (slice == 0) {
      float* grow = IO.dgate_logits + ((size_t)b * L + i) * L;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j0 + r < L) atomicAdd(grow + j0 + r, dgl[r]);
    }
  } else if (IO.dgate_logits) {
    float* gbase = IO.dgate_logits + bh * (size_t)L * L;
    for (int row = 0; row < L; ++row) {
      if (row == i) continue;
      for (int j = lane; j < L; j += 64) gbase[(size_t)row * L + j] = 0.f;
    }
    if (slice == 0) store_seg(gbase + (size_t)i * L, j0, L, true, dgl);
  }
  __syncthreads();  // col[] complete
  OR_STAMP(4);

  // ---- query side and key halves of the parameter gradients: lane d sums over the keys --------------------------------------
  const int stride_w = IO.part_stride ? IO.part_stride : 2 * DH, stride_s = IO.part_stride ? IO.part_stride : 4;
  // lane = (column d, part): with DH <= 32 the 64 / DH parts of a column split the keys (and the rows to zero) among them
  // (head size 128 [r4]: two column halves, one after the other -- round 3's form wrote columns 0..63 only, and its tests
  // read what an earlier launch had left in the reused buffer)
  constexpr int PARTS = DH <= 32 ? 64 / DH : 1;
  constexpr int NCH = DH > 64 ? DH / 64 : 1;
#pragma unroll 1
  for (int ch = 0; ch < NCH; ++ch) {
    const int d = (DH > 64 ? lane : lane % DH) + 64 * ch, part = DH > 64 ? 0 : lane / DH;
    float sq = 0.f, sqa = 0.f, so = 0.f, sd = 0.f;
    constexpr int UJ = 8;  // keys per trip: their 16 loads are requested before the first is used
    for (int jb = part; jb < L; jb += PARTS * UJ) {
      float kv[UJ], kav[UJ];
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        const int j = min(jb + u * PARTS, L - 1);
        const size_t o = (rowbase + j) * H + hoff + d;
        kv[u] = P.k[o];
        kav[u] = P.ka[o];
      }
#pragma unroll
      for (int u = 0; u < UJ; ++u) {
        const int j = jb + u * PARTS;
        const float w = j < L ? 1.0f : 0.0f;
        const int jc = min(j, L - 1);
        sq += w * col[0][jc] * kv[u];
        sqa += w * col[1][jc] * kav[u];
        so += w * col[2][jc] * kv[u];
        sd += w * col[3][jc] * kv[u];
      }
    }
#pragma unroll
    for (int off = DH; off < 64; off <<= 1) {
      sq += __shfl_xor(sq, off);
      sqa += __shfl_xor(sqa, off);
      so += __shfl_xor(so, off);
      sd += __shfl_xor(sd, off);
    }
    // every row of dq / dqa but the read one is zero: 16-byte stores, 64 / (DH / 4) rows per instruction ([r4]: one dword per
    // lane and row was 2 L / PARTS store instructions of 256 bytes -- 200 of them at L = 200)
    if (ch == 0) {
      constexpr int LPR = DH / 4, RPI = 64 / LPR;  // lanes per row slice, rows per instruction
      const f4 z4 = {0.f, 0.f, 0.f, 0.f};
      for (int row = lane / LPR; row < L; row += RPI) {
        if (row == i) continue;
        const size_t o = (rowbase + row) * H + hoff + 4 * (lane % LPR);
        *(f4*)(IO.dq + o) = z4;
        if (!ACC) *(f4*)(IO.dqa + o) = z4;
      }
    }
    if (part == 0) {
      const size_t o = (rowbase + i) * H + hoff + d;
      IO.dq[o] = sq + da_o * vec[6][d] + da_d * vec[7][d];
      if (ACC) IO.dqa[o] += sqa;
      else IO.dqa[o] = sqa;
    }
    if (part != 0) continue;
    IO.dw_order_part[bh * stride_w + d] = da_o * vec[0][d];
    IO.dw_dist_part[bh * stride_w + d] = da_d * vec[0][d];
    IO.dw_order_part[bh * stride_w + DH + d] = so;
    IO.dw_dist_part[bh * stride_w + DH + d] = sd;
  }
  if (lane == 0) {
    float* sm = IO.dsmall_part + bh * stride_s;
    sm[0] = da_o;
    sm[1] = da_d;
    sm[2] = dsc_sum;
    sm[3] = 0.f;
  }
  OR_STAMP(5);
#ifdef ACATTN_ONEROW_STAMPS
  if (lane == 0 && blockIdx.x < 2048)
    for (int k = 0; k < 8; ++k) g_onerow_stamps[blockIdx.x * 8 + k] = cyc_[k];
#endif
}

#ifdef ACATTN_ONEROW_STAMPS
extern "C" int acattn_debug_onerow_stamps(unsigned long long* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_onerow_stamps), (size_t)n_words * 8);
}
#endif

// Returns -100 when the one-row form does not apply.
template <int DH>
int launch_onerow(const acattn_problem& p, const acattn_bwd_io& io, bool accumulate, hipStream_t stream) {
  const dim3 grid(p.B * p.n_heads), block(64);
  if (io.dgate_logits && io.dgate_summed) {  // the heads add their read row into a zeroed [B,L,L]
    const int rc = acattn_launch_zero(io.dgate_logits, (size_t)p.B * p.L * p.L, stream);
    if (rc) return rc;
  }
  if (p.L <= 64) {
    if (accumulate)
      hipLaunchKernelGGL((acattn_bwd_onerow_kernel<DH, true, true>), grid, block, 0, stream, p, io);
    else
      hipLaunchKernelGGL((acattn_bwd_onerow_kernel<DH, false, true>), grid, block, 0, stream, p, io);
  } else if (accumulate) {
    hipLaunchKernelGGL((acattn_bwd_onerow_kernel<DH, true, false>), grid, block, 0, stream, p, io);
  } else {
    hipLaunchKernelGGL((acattn_bwd_onerow_kernel<DH, false, false>), grid, block, 0, stream, p, io);
  }
  return (int)hipGetLastError();
}

template <int DH>
int launch_stream(const acattn_problem& p, const acattn_bwd_io& io, float* ws, hipStream_t stream) {
  const int nT = (p.L + 15) / 16, dh = DH;
  const size_t rows = (size_t)p.B * p.n_heads;
  if (!io.attack_only) {  // the parameter partial rows are accumulated with atomics: start from zero
    // (acattn_launch_zero, not hipMemsetAsync: see acattn_util.hip)
    // a failed fill must not be followed by atomics into whatever the buffer held
    int rc = 0;
    auto zero = [&](float* ptr, size_t n) { if (!rc) rc = acattn_launch_zero(ptr, n, stream); };
    if (io.part_stride) {
      float* lo = std::min(io.dw_order_part, std::min(io.dw_dist_part, io.dsmall_part));
      zero(lo, rows * io.part_stride);
    } else {
      zero(io.dw_order_part, rows * 2 * dh);
      zero(io.dw_dist_part, rows * 2 * dh);
      zero(io.dsmall_part, rows * 4);
    }
    if (rc) return rc;
  }
  const dim3 grid(p.B * p.n_heads * nT), block(64);
  static const bool no_dca_form = getenv("ACATTN_BWD_DCA_ALWAYS") ? atoi(getenv("ACATTN_BWD_DCA_ALWAYS")) == 0 : true;
  if (io.d_ctx_attacked || !no_dca_form) {
    hipLaunchKernelGGL((acattn_bwd_row_kernel<DH, true>), grid, block, 0, stream, p, io, ws);
    hipLaunchKernelGGL((acattn_bwd_key_kernel<DH, true>), grid, block, 0, stream, p, io, (const float*)ws);
  } else if (io.dqa2) {  // [r4] ... and a second cotangent set riding on the same rebuilt tiles (second_set_dsa)
    hipLaunchKernelGGL((acattn_bwd_row_kernel<DH, false, true>), grid, block, 0, stream, p, io, ws);
    hipLaunchKernelGGL((acattn_bwd_key_kernel<DH, false, true>), grid, block, 0, stream, p, io, (const float*)ws);
  } else {  // [r4] no cotangent of the attacked context: no perturbed attention, no Gaussian noise (see tile_elementwise)
    hipLaunchKernelGGL((acattn_bwd_row_kernel<DH, false>), grid, block, 0, stream, p, io, ws);
    hipLaunchKernelGGL((acattn_bwd_key_kernel<DH, false>), grid, block, 0, stream, p, io, (const float*)ws);
  }
  return (int)hipGetLastError();
}

}  // namespace

int64_t acattn_bwd_stream_ws_bytes(const acattn_problem& p) { return (int64_t)p.B * p.n_heads * p.L * NSC * sizeof(float); }

// The calibrated-loss pass through the LAST layer: only the read position of every sequence carries a cotangent
// (io.read_rows with one position per sequence, no mask cotangent).  Returns -100 when that is not the situation.
// `accumulate`: see the kernel (the caller has run the mask path; io.d_attack_mask must then be NULL here).
bool acattn_bwd_onerow_applies(const acattn_problem& p, const acattn_bwd_io& io) {
  static const bool enabled = getenv("ACATTN_ONEROW") ? atoi(getenv("ACATTN_ONEROW")) != 0 : true;
  const int dh = p.n_heads > 0 ? p.H / p.n_heads : 0;
  return enabled && io.read_rows && io.n_read_rows == 1 && !io.active_qblocks && !io.d_attack_mask && !io.d_penalty_part && !io.attack_only &&
         p.L <= 208 && (int64_t)p.B * p.n_heads * p.L * p.L < (1LL << 30) && (int64_t)p.B * p.L * p.H < (1LL << 30) &&
         p.mask_mode == ACATTN_MASK_STRUCTURED && p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist &&
         p.adversarial && p.combine_option == ACATTN_COMBINE_GATE && p.two_level && (dh == 16 || dh == 32 || dh == 64 || dh == 128);
}

int acattn_launch_bwd_onerow(const acattn_problem& p, const acattn_bwd_io& io, bool accumulate, hipStream_t stream) {
  if (!acattn_bwd_onerow_applies(p, io)) return -100;
  switch (p.H / p.n_heads) {
    case 16: return launch_onerow<16>(p, io, accumulate, stream);
    case 32: return launch_onerow<32>(p, io, accumulate, stream);
    case 64: return launch_onerow<64>(p, io, accumulate, stream);
    case 128: return launch_onerow<128>(p, io, accumulate, stream);  // [r3]
  }
  return -100;
}

// acattn_bwd_io's second cotangent set (d_ctx_calibrated2 / d_penalty_part2 -> dqa2, dka2): the streaming pair's form without
// an attacked-context cotangent, every query block on the full path (no read_rows / active_qblocks hints), dh <= 64 (at head
// size 128 the key kernel has no registers left for it).
bool acattn_bwd_stream_pair_applies(const acattn_problem& p, const acattn_bwd_io& io) {
  const int dh = p.n_heads > 0 ? p.H / p.n_heads : 0;
  return io.workspace && p.L <= 208 && (int64_t)p.B * p.n_heads * p.L * p.L < (1LL << 30) && (int64_t)p.B * p.L * p.H < (1LL << 30) &&
         p.mask_mode == ACATTN_MASK_STRUCTURED && p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist && p.adversarial &&
         p.combine_option == ACATTN_COMBINE_GATE && p.two_level && (dh == 16 || dh == 32 || dh == 64) && !io.d_ctx_attacked &&
         !io.attack_only && !io.read_rows && !io.active_qblocks && io.dqa2 && io.dka2 &&
         (io.d_ctx_calibrated2 || io.d_penalty_part2);
}

// Returns -100 when the problem is outside this path's domain (the caller then uses the row-resident kernels).
int acattn_launch_bwd_stream(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  const bool ok = io.workspace && p.L <= 208 && (int64_t)p.B * p.n_heads * p.L * p.L < (1LL << 30) &&
                  (int64_t)p.B * p.L * p.H < (1LL << 30) && p.mask_mode == ACATTN_MASK_STRUCTURED &&
                  p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist && p.adversarial &&
                  p.combine_option == ACATTN_COMBINE_GATE && p.two_level;
  if (!ok) return -100;
  float* ws = (float*)io.workspace;
  switch (p.H / p.n_heads) {
    case 16: return launch_stream<16>(p, io, ws, stream);
    case 32: return launch_stream<32>(p, io, ws, stream);
    case 64: return launch_stream<64>(p, io, ws, stream);
    case 128: return launch_stream<128>(p, io, ws, stream);  // [r3]
  }
  return -100;
}
