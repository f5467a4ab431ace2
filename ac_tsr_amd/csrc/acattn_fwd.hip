// Fused calibrated multi-head self-attention, forward, for gfx950 (MI355X).
//
// One workgroup = one (sequence b, head h).  Each wave owns 16-row query blocks of that head and
// keeps whole attention rows in registers in "key-major" MFMA layout: the tile S^T = K . Q^T is
// computed with v_mfma_f32_16x16x4_f32, so lane (c = lane&15, g = lane>>4) holds, for query row
// i0 + c, the keys 16t + 4g + r (t = key tile, r = accumulator register).  In that layout
//   * every row softmax is a register-local reduction plus two permlane swaps (quad_sum/quad_max),
//   * the probability registers are directly the B operand of the P.V product
//     (ctx^T = V^T . P^T), so probabilities never touch LDS or HBM,
//   * gate logits, noise and the attack mask M move as 16-byte row segments.
// V of the head is staged once per workgroup in LDS (padded rows, conflict-free ds_read_b32);
// K/Ka/Q/Qa fragments are loaded straight from HBM/L2 in MFMA operand order.
// The spatial calibrator's affine over the concatenation (q_i || k_j) is evaluated in its rank-1
// form a_i + c_j + bias (reference materialises [B,h,L,L,2dh]: recbole/model/layers.py:705-708).
//
// Reference semantics reproduced (recbole/model/layers.py): :695-740 scores, spatial calibrator and
// the two first-level softmaxes; :661-672 attack mask; :917-936 adversarial calibrator; :677-680 P.V.
#include <type_traits>

#include <stdlib.h>

#include "acattn_common.h"

namespace {

// FAST = the training configuration of every shipped reference config, fixed at compile time:
// structured causal/bidirectional mask, counter RNG, gate combine, both spatial terms, two_level.
// Every other combination runs the same code with the flags read at run time.
template <int DH, int NT, bool ADV, bool FULL, bool FAST>
__global__ void __launch_bounds__(256) acattn_fwd_kernel(const acattn_problem P, const acattn_fwd_out O) {
  constexpr int KS = DH / 4;   // k-steps of the score MFMAs == floats of a row fragment held per lane
  constexpr int DT = DH / 16;  // 16-wide d tiles of the context
  constexpr int VS = DH + 4;   // LDS row stride of V (stride % 8 == 4 -> ds_read_b32 conflict-free)

  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;  // key tiles == query blocks
  const int LP = nT * 16;
  int b, h;
  decode_block(blockIdx.x, P.B, nh, b, h);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * L;  // first row of this sequence in a [B*L, H] tensor
  const int hoff = h * DH;
  const size_t bh = (size_t)b * nh + h;
  const bool structured = FAST ? true : P.mask_mode == ACATTN_MASK_STRUCTURED;
  const bool use_order = FAST ? true : P.w_order != nullptr, use_dist = FAST ? true : P.w_dist != nullptr;
  const int combine_option = FAST ? (int)ACATTN_COMBINE_GATE : P.combine_option;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Vs = smem;            // [LP][VS]
  float* s_co = Vs + LP * VS;  // key-side half of order_affine:    k_j . w_order[dh:]
  float* s_cd = s_co + LP;     // key-side half of distance_affine: k_j . w_dist[dh:]
  float* s_km = s_cd + LP;     // per-key additive mask: 0 / -10000 / (-inf for j >= L)
  float* s_lt = s_km + LP;     // log(d + 1), d = 0..LP-1   (layers.py:721-723)

  // ---- stage V, key-side calibrator terms, key mask -------------------------------------------
  for (int idx = threadIdx.x; idx < LP * (DH / 4); idx += blockDim.x) {
    const int row = idx / (DH / 4), c4 = idx - row * (DH / 4);
    f4 val = {0.f, 0.f, 0.f, 0.f};
    if (row < L) val = *(const f4*)(P.v + (rowbase + row) * H + hoff + 4 * c4);
    *(f4*)(Vs + row * VS + 4 * c4) = val;
  }
  for (int idx = threadIdx.x; idx < LP * 4; idx += blockDim.x) {
    // 4 adjacent lanes share one key: each takes a quarter of the head dimension
    const int j = idx >> 2, part = idx & 3;
    float co = 0.f, cd = 0.f;
    if (j < L && (use_order || use_dist)) {
      const float* kr = P.k + (rowbase + j) * H + hoff + part * (DH / 4);
#pragma unroll
      for (int d4 = 0; d4 < DH / 16; ++d4) {
        const f4 kv = *(const f4*)(kr + 4 * d4);
        if (use_order) {
          const f4 w = *(const f4*)(P.w_order + DH + part * (DH / 4) + 4 * d4);
          co += kv.x * w.x + kv.y * w.y + kv.z * w.z + kv.w * w.w;
        }
        if (use_dist) {
          const f4 w = *(const f4*)(P.w_dist + DH + part * (DH / 4) + 4 * d4);
          cd += kv.x * w.x + kv.y * w.y + kv.z * w.z + kv.w * w.w;
        }
      }
    }
    co += __shfl_xor(co, 1);
    co += __shfl_xor(co, 2);
    cd += __shfl_xor(cd, 1);
    cd += __shfl_xor(cd, 2);
    if (part == 0) {
      float km = ACATTN_NEG_INF;
      if (j < L) {
        if (structured)
          km = P.key_valid[rowbase + j] ? 0.f : ACATTN_MASK_FILL;
        else if (P.mask_mode == ACATTN_MASK_DENSE_L)
          km = P.mask[rowbase + j];
        else
          km = 0.f;
      }
      s_co[j] = co;
      s_cd[j] = cd;
      s_km[j] = km;
      s_lt[j] = logf((float)(j + 1));
    }
  }
  __syncthreads();

  // first unmasked key of the sequence (structured masks): rows before it are fully masked and
  // then spread over ALL keys, so causal tile skipping is only legal from that row on.
  int first_valid = L, last_valid = -1;
  if (structured) {
    for (int j = lane; j < L; j += 64)
      if (s_km[j] == 0.f) {
        first_valid = min(first_valid, j);
        last_valid = max(last_valid, j);
      }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      first_valid = min(first_valid, __shfl_xor(first_valid, off));
      last_valid = max(last_valid, __shfl_xor(last_valid, off));
    }
  }
  const int nt_valid = last_valid >= 0 ? (last_valid >> 4) + 1 : nT;  // tiles past the last real item hold no mass

  // ---- per-lane constants -----------------------------------------------------------------------
  float wo_q[KS], wd_q[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    wo_q[s] = use_order ? P.w_order[KS * g + s] : 0.f;
    wd_q[s] = use_dist ? P.w_dist[KS * g + s] : 0.f;
  }
  const float b_o = use_order ? P.b_order[0] : 0.f;
  const float b_d = use_dist ? P.b_dist[0] : 0.f;
  const float sc = use_dist ? P.scalar[0] : 0.f;
  const float s2 = sc * sc;
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const bool has_drop = P.p_drop > 0.f;
  const float keep_scale = has_drop ? 1.0f / (1.0f - P.p_drop) : 1.0f;
  const bool counter = FAST ? true : P.rng_mode == ACATTN_RNG_COUNTER;
  const uint64_t seed_eff = P.seed + (P.seed_device ? *P.seed_device : 0ull);
  const RngKey rkey = rng_key(seed_eff);

  for (int kk = 0;; ++kk) {
    const int qb = (kk & 1) ? (kk + 1) * NW - 1 - wave : kk * NW + wave;  // zig-zag: balances causal work
    if (qb >= nT) break;
    const int i0 = qb * 16, i = i0 + c;
    const bool row_ok = i < L;
    int nt = nT;
    if (structured && (P.causal ? first_valid <= i0 : last_valid >= 0))
      nt = min(P.causal ? min(nT, qb + 1) : nT, nt_valid);
    const size_t prow = (bh * L + (row_ok ? i : 0)) * (size_t)L;  // row offset into [B,nh,L,L] tensors

    // The block body is instantiated once per number of processed key tiles (NTB = 1..NT) when NT <= 4:
    // straight-line code with no per-tile branches, so every global load of the block can be issued
    // before the first MFMA.  NTB == 0 keeps wave-uniform `t < nt` guards (long sequences).
    auto body = [&](auto ntb_c) {
    constexpr int NTB = decltype(ntb_c)::value;
#define TILE_ON(t) (NTB ? ((t) < NTB) : ((t) < nt))

    // ---- query fragments and query-side calibrator terms ----------------------------------------
    float qf[KS], qaf[KS];
    {
      const float* qp = P.q + (rowbase + i) * H + hoff + KS * g;
      const float* qap = ADV ? P.qa + (rowbase + i) * H + hoff + KS * g : nullptr;
#pragma unroll
      for (int s4 = 0; s4 < KS / 4; ++s4) {
        f4 t = {0.f, 0.f, 0.f, 0.f}, ta = {0.f, 0.f, 0.f, 0.f};
        if (row_ok) {
          t = *(const f4*)(qp + 4 * s4);
          if (ADV) ta = *(const f4*)(qap + 4 * s4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qf[4 * s4 + e] = t[e];
          qaf[4 * s4 + e] = ta[e];
        }
      }
    }
    float ao = 0.f, ad = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      ao += qf[s] * wo_q[s];
      ad += qf[s] * wd_q[s];
    }
    ao = quad_sum(ao) + b_o;
    ad = quad_sum(ad) + b_d;

    // ---- additive mask of this row block ---------------------------------------------------------
    float mk[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (TILE_ON(t)) {
        const f4 km4 = *(const f4*)(s_km + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * t + 4 * g + r;
          float m = km4[r];
          if (structured) {
            if (P.causal && j > i) m = fminf(m, ACATTN_MASK_FILL);
          } else if (P.mask_mode == ACATTN_MASK_DENSE_LL) {
            if (j < L && row_ok) m = P.mask[(rowbase + i) * L + j];
          }
          mk[t][r] = m;
        }
      }
    }

    // ---- S^T = K.Q^T and Sa^T = Ka.Qa^T on the matrix cores -------------------------------------
    f4 accS[NT], accM[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      accS[t] = f4{0.f, 0.f, 0.f, 0.f};
      accM[t] = f4{0.f, 0.f, 0.f, 0.f};
      if (TILE_ON(t)) {
        const int j = 16 * t + c;
        const bool kok = j < L;
        const size_t koff = (rowbase + (kok ? j : 0)) * H + hoff + KS * g;
        float kf[KS], kaf[KS];
#pragma unroll
        for (int s4 = 0; s4 < KS / 4; ++s4) {
          f4 t4 = {0.f, 0.f, 0.f, 0.f}, ta4 = {0.f, 0.f, 0.f, 0.f};
          if (kok) {
            t4 = *(const f4*)(P.k + koff + 4 * s4);
            if (ADV) ta4 = *(const f4*)(P.ka + koff + 4 * s4);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            kf[4 * s4 + e] = t4[e];
            kaf[4 * s4 + e] = ta4[e];
          }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          accS[t] = mfma16(kf[s], qf[s], accS[t]);
          if (ADV) accM[t] = mfma16(kaf[s], qaf[s], accM[t]);
        }
      }
    }

    // ---- first-level softmaxes: after_spatial P (and before_spatial P0 when asked) --------------
    f4 accB[FULL ? NT : 1];  // raw scores for before_spatial
    if (FULL) {
#pragma unroll
      for (int t = 0; t < NT; ++t) accB[FULL ? t : 0] = accS[t];
    }
    float mx = ACATTN_NEG_INF, row_shift = ACATTN_NEG_INF;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (TILE_ON(t)) {
        const f4 co4 = *(const f4*)(s_co + 16 * t + 4 * g);
        const f4 cd4 = *(const f4*)(s_cd + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * t + 4 * g + r;
          float s = accS[t][r];
          if (use_order) {  // layers.py:715-719
            const float pr = fast_sigmoid(ao + co4[r]);
            const float val = (j > i) ? pr : 1.0f - pr;
            s += fast_log(val + ACATTN_LOG_EPS);
          }
          if (use_dist) {  // layers.py:721-727
            const int dist = i > j ? i - j : j - i;
            const float df = s_lt[dist] - (ad + cd4[r]);
            s += -0.5f * ((df * df) * s2);
          }
          const float x = s * inv_sqrt + mk[t][r];  // layers.py:732-734
          accS[t][r] = x;
          mx = fmaxf(mx, x);
          row_shift = fmaxf(row_shift, mk[t][r]);
        }
      }
    }
    mx = quad_max(mx);
    row_shift = quad_max(row_shift);  // 0 when the row has an unmasked key, else -10000
    float zx = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (TILE_ON(t)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = fast_exp(accS[t][r] - mx);
          accS[t][r] = e;
          zx += e;
        }
      }
    }
    zx = quad_sum(zx);
    float stat[ACATTN_NSTAT];
#pragma unroll
    for (int s = 0; s < ACATTN_NSTAT; ++s) stat[s] = 0.f;
    stat[0] = mx + fast_log(zx);

    // randomness of this row block
    float nz[NT][4];
    uint32_t keepA = 0xFFFFFFFFu, keepM = 0xFFFFFFFFu, keepB = 0xFFFFFFFFu;  // bit 4t+r (NT <= 8) -- see below
    uint32_t keepA2 = 0xFFFFFFFFu, keepM2 = 0xFFFFFFFFu, keepB2 = 0xFFFFFFFFu;  // tiles 8..15
    if (ADV || has_drop) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
          const int j0 = 16 * t + 4 * g;
          uint32_t ka = 0xFu, km_ = 0xFu, kb = 0xFu;
          if (counter) {
            const RngGroup rg = rng_group(rkey, (uint32_t)(bh * L + i), (uint32_t)(4 * t + g), P.p_drop);
#pragma unroll
            for (int r = 0; r < 4; ++r) nz[t][r] = rg.n[r];
            if (has_drop) {
              ka = rg.keep_after;
              km_ = rg.keep_mask;
              kb = rg.keep_before;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = j0 + r;
              const bool ok = row_ok && j < L;
              nz[t][r] = (ADV && ok && P.noise) ? P.noise[prow + j] : 0.f;
              if (has_drop && ok) {
                if (P.keep_after && !P.keep_after[prow + j]) ka &= ~(1u << r);
                if (ADV && P.keep_mask && !P.keep_mask[prow + j]) km_ &= ~(1u << r);
                if (FULL && P.keep_before && !P.keep_before[prow + j]) kb &= ~(1u << r);
              }
            }
          }
          if (t < 8) {
            keepA = (keepA & ~(0xFu << (4 * t))) | (ka << (4 * t));
            keepM = (keepM & ~(0xFu << (4 * t))) | (km_ << (4 * t));
            keepB = (keepB & ~(0xFu << (4 * t))) | (kb << (4 * t));
          } else {
            keepA2 = (keepA2 & ~(0xFu << (4 * (t - 8)))) | (ka << (4 * (t - 8)));
            keepM2 = (keepM2 & ~(0xFu << (4 * (t - 8)))) | (km_ << (4 * (t - 8)));
            keepB2 = (keepB2 & ~(0xFu << (4 * (t - 8)))) | (kb << (4 * (t - 8)));
          }
        }
      }
    }
    auto kept = [&](uint32_t lo, uint32_t hi, int t, int r) -> bool {
      return t < 8 ? ((lo >> (4 * t + r)) & 1u) : ((hi >> (4 * (t - 8) + r)) & 1u);
    };

    {  // P = dropout(softmax)   layers.py:735-736
      const float rz = fast_rcp(zx) * keep_scale;
      const float rz0 = fast_rcp(zx);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            accS[t][r] = has_drop ? (kept(keepA, keepA2, t, r) ? accS[t][r] * rz : 0.f) : accS[t][r] * rz0;
        }
      }
    }

    // helpers for [B,nh,L,L] row-segment stores (16 B when the 4 keys exist, scalar at the ragged end)
    auto store_seg = [&](float* base, int t, const f4 val) {
      const int j0 = 16 * t + 4 * g;
      if (!row_ok || j0 >= L) return;
      float* p = base + prow + j0;
      if (j0 + 3 < L) {
        *(f4u*)p = val;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + r < L) p[r] = val[r];
      }
    };
    auto zero_skipped = [&](float* base) {
      for (int t = nt; t < nT; ++t) store_seg(base, t, f4{0.f, 0.f, 0.f, 0.f});
    };

    f4 after[FULL ? NT : 1];  // after_spatial kept for the one-level combine (layers.py:929-934)
    if (FULL) {
#pragma unroll
      for (int t = 0; t < NT; ++t) after[FULL ? t : 0] = accS[t];
      if (O.after_spatial) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (TILE_ON(t)) store_seg(O.after_spatial, t, accS[t]);
        zero_skipped(O.after_spatial);
      }
      // before_spatial = dropout(softmax(raw / sqrt(dh) + mask))   layers.py:740
      float mb = ACATTN_NEG_INF;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float x = accB[FULL ? t : 0][r] * inv_sqrt + mk[t][r];
            accB[FULL ? t : 0][r] = x;
            mb = fmaxf(mb, x);
          }
        }
      }
      mb = quad_max(mb);
      float zb = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = fast_exp(accB[FULL ? t : 0][r] - mb);
            accB[FULL ? t : 0][r] = e;
            zb += e;
          }
        }
      }
      zb = quad_sum(zb);
      stat[6] = mb + fast_log(zb);
      const float rzb = fast_rcp(zb);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pb = accB[FULL ? t : 0][r] * rzb;
            if (has_drop) pb = kept(keepB, keepB2, t, r) ? pb * keep_scale : 0.f;
            accB[FULL ? t : 0][r] = pb;
          }
          if (O.before_spatial) store_seg(O.before_spatial, t, accB[FULL ? t : 0]);
        }
      }
      if (O.before_spatial) zero_skipped(O.before_spatial);
      if (!P.two_level) {  // origin = before_spatial   layers.py:913-914
#pragma unroll
        for (int t = 0; t < NT; ++t) accS[t] = accB[FULL ? t : 0];
      }
    }

    f4 cc[DT], ca[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      cc[dt] = f4{0.f, 0.f, 0.f, 0.f};
      ca[dt] = f4{0.f, 0.f, 0.f, 0.f};
    }

    if (!ADV) {
      // spatial calibrator only: ctx = after_spatial . V
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* vp = Vs + (16 * t + 4 * g + r) * VS + c;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) cc[dt] = mfma16(vp[16 * dt], accS[t][r], cc[dt]);
          }
        }
      }
    } else {
      // ---- attack mask M = dropout(softmax(Sa / sqrt(dh) + mask))   layers.py:664-672 -------------
      float my = ACATTN_NEG_INF;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float y = accM[t][r] * inv_sqrt + mk[t][r];
            accM[t][r] = y;
            my = fmaxf(my, y);
          }
        }
      }
      my = quad_max(my);
      float zy = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = fast_exp(accM[t][r] - my);
            accM[t][r] = e;
            zy += e;
          }
        }
      }
      zy = quad_sum(zy);
      stat[1] = my + fast_log(zy);
      {
        const float rz = fast_rcp(zy);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (TILE_ON(t)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float m = accM[t][r] * rz;
              if (has_drop) m = kept(keepM, keepM2, t, r) ? m * keep_scale : 0.f;
              accM[t][r] = m;
            }
            store_seg(O.attack_mask, t, accM[t]);
          }
        }
        zero_skipped(O.attack_mask);
      }

      // ---- perturbed attention: softmax(P*M + n*(1-M) + mask)   layers.py:918-919 ---------------
      f4 eu[NT];
      float mu = ACATTN_NEG_INF;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = accS[t][r], m = accM[t][r];
            const float u = (p * m + nz[t][r] * (1.0f - m)) + mk[t][r];
            eu[t][r] = u;
            mu = fmaxf(mu, u);
          }
        }
      }
      mu = quad_max(mu);
      float zu = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = fast_exp(eu[t][r] - mu);
            eu[t][r] = e;
            zu += e;
          }
        }
      }
      zu = quad_sum(zu);
      stat[2] = mu + fast_log(zu);
      const float rzu = fast_rcp(zu);
      if (FULL && O.perturbed_attention) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (TILE_ON(t)) store_seg(O.perturbed_attention, t, eu[t] * rzu);
        zero_skipped(O.perturbed_attention);
      }
      // attacked context: (sum_j e_u V_j) / Z_u
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* vp = Vs + (16 * t + 4 * g + r) * VS + c;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) ca[dt] = mfma16(vp[16 * dt], eu[t][r], ca[dt]);
          }
        }
      }
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) ca[dt] *= rzu;

      // ---- calibrated attention: softmax(P*exp(1-M) + mask)   layers.py:920-921 -----------------
      f4 ev[NT];
      float zv = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float v_ = accS[t][r] * fast_exp(1.0f - accM[t][r]) + mk[t][r];
            const float e = fast_exp(v_ - row_shift);  // inputs lie in [0, e] + mask: shift by the row's mask max
            ev[t][r] = e;
            zv += e;
          }
        }
      }
      zv = quad_sum(zv);
      stat[3] = row_shift + fast_log(zv);
      const float rzv = fast_rcp(zv);

      // ---- combine (layers.py:883-896) and final softmax (:925) ---------------------------------
      float zf = 0.f;
      if (combine_option == ACATTN_COMBINE_FIXED) {
        // inner softmax(P + 0.5*A_c) carries NO mask: every existing key takes part, the skipped
        // (all-zero) tiles contribute exp(0) each.
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (TILE_ON(t)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = 16 * t + 4 * g + r;
              const float e = (j < L) ? fast_exp(accS[t][r] + 0.5f * (ev[t][r] * rzv)) : 0.f;
              ev[t][r] = e;
              zf += e;
            }
          }
        }
        zf = quad_sum(zf) + (float)(L - min(L, 16 * nt));
        stat[5] = fast_log(zf);
        const float rzf = fast_rcp(zf);
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (TILE_ON(t)) ev[t] *= rzf;
      } else if (combine_option == ACATTN_COMBINE_GATE) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (TILE_ON(t)) {
            const int j0 = 16 * t + 4 * g;
            f4 gl = {0.f, 0.f, 0.f, 0.f};
            if (row_ok && j0 < L) {
              const float* gp = P.gate_logits + (rowbase + i) * L + j0;
              if (j0 + 3 < L) {
                gl = *(const f4u*)gp;
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (j0 + r < L) gl[r] = gp[r];
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float gt = fast_sigmoid(gl[r]);
              ev[t][r] = gt * accS[t][r] + (1.0f - gt) * (ev[t][r] * rzv);
            }
          }
        }
      } else {  // annealing
        const float rate = P.anneal_rate;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          if (TILE_ON(t)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ev[t][r] = rate * accS[t][r] + (1.0f - rate) * (ev[t][r] * rzv);
          }
        }
      }
      float zw = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float e = fast_exp((ev[t][r] + mk[t][r]) - row_shift);
            ev[t][r] = e;
            zw += e;
          }
        }
      }
      zw = quad_sum(zw);
      stat[4] = row_shift + fast_log(zw);
      const float rzw = fast_rcp(zw);
      bool normalised = false;
      if (FULL) {
        if (O.calibrated_attention || !P.two_level) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (TILE_ON(t)) ev[t] *= rzw;
          normalised = true;
          if (O.calibrated_attention) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
              if (TILE_ON(t)) store_seg(O.calibrated_attention, t, ev[t]);
            zero_skipped(O.calibrated_attention);
          }
        }
        if (!P.two_level) {  // layers.py:929-934
          const float ratio = P.rich_combine == ACATTN_RICH_TRAINABLE ? P.rich_ratio[0] : 0.5f;
#pragma unroll
          for (int t = 0; t < NT; ++t) {
            if (TILE_ON(t)) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (P.rich_combine == ACATTN_RICH_TRAINABLE)
                  ev[t][r] = ratio * ev[t][r] + (1.0f - ratio) * after[FULL ? t : 0][r];
                else
                  ev[t][r] = (ev[t][r] + after[FULL ? t : 0][r]) / 2.0f;
              }
            }
          }
        }
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (TILE_ON(t)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float* vp = Vs + (16 * t + 4 * g + r) * VS + c;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) cc[dt] = mfma16(vp[16 * dt], ev[t][r], cc[dt]);
          }
        }
      }
      if (!normalised) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) cc[dt] *= rzw;
      }
    }

    // ---- write contexts (head-merged [B,L,H]) and the row statistics ------------------------------
    if (row_ok) {
      float* oc = O.ctx_calibrated + (rowbase + i) * H + hoff + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) *(f4*)(oc + 16 * dt) = cc[dt];
      if (ADV) {
        float* oa = O.ctx_attacked + (rowbase + i) * H + hoff + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *(f4*)(oa + 16 * dt) = ca[dt];
      }
      if (O.row_stats && g == 0) {
        float* sp = O.row_stats + (bh * L + i) * ACATTN_NSTAT;
        *(f4*)sp = f4{stat[0], stat[1], stat[2], stat[3]};
        *(f4*)(sp + 4) = f4{stat[4], stat[5], stat[6], stat[7]};
      }
    }
#undef TILE_ON
    };  // body
    if constexpr (NT <= 4) {
      switch (nt) {
        case 1: body(std::integral_constant<int, 1>{}); break;
        case 2: body(std::integral_constant<int, NT >= 2 ? 2 : 1>{}); break;
        case 3: body(std::integral_constant<int, NT >= 3 ? 3 : 1>{}); break;
        default: body(std::integral_constant<int, NT>{}); break;
      }
    } else {
      body(std::integral_constant<int, 0>{});
    }
  }
}

__global__ void acattn_rng_kernel(int B, int nh, int L, uint64_t seed, float p_drop, float* noise, uint8_t* keep_after,
                                  uint8_t* keep_mask, uint8_t* keep_before) {
  const int groups = (L + 3) / 4;
  const size_t total = (size_t)B * nh * L * groups;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t row = (uint32_t)(idx / groups), grp = (uint32_t)(idx % groups);
    const RngGroup rg = rng_group(rng_key(seed), row, grp, p_drop);
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * grp + r;
      if (j >= L) break;
      const size_t o = (size_t)row * L + j;
      if (noise) noise[o] = rg.n[r];
      if (keep_after) keep_after[o] = (rg.keep_after >> r) & 1u;
      if (keep_mask) keep_mask[o] = (rg.keep_mask >> r) & 1u;
      if (keep_before) keep_before[o] = (rg.keep_before >> r) & 1u;
    }
  }
}

template <int DH, int NT>
int launch_nt(const acattn_problem& p, const acattn_fwd_out& o, bool full, hipStream_t stream) {
  const int nT = (p.L + 15) / 16;
  const int NW = nT <= 4 ? nT : 4;  // one wave per 16-row query block while they fit; zig-zag beyond
  const int LP = nT * 16;
  const size_t lds = (size_t)(LP * (DH + 4) + 4 * LP) * sizeof(float);
  const dim3 grid(p.B * p.n_heads), block(64 * NW);
  const bool fast = p.mask_mode == ACATTN_MASK_STRUCTURED && p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist &&
                    (!p.adversarial || (p.combine_option == ACATTN_COMBINE_GATE && p.two_level));
  auto launch = [&](auto kern) {
    // head size 128 at L > 96 needs more than the default 64 KB of dynamic LDS
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p, o);
  };
  if (!p.adversarial) {
    if (fast)
      launch(acattn_fwd_kernel<DH, NT, false, false, true>);
    else
      launch(acattn_fwd_kernel<DH, NT, false, false, false>);
  } else if (full) {
    launch(acattn_fwd_kernel<DH, NT, true, true, false>);
  } else if (fast) {
    launch(acattn_fwd_kernel<DH, NT, true, false, true>);
  } else {
    launch(acattn_fwd_kernel<DH, NT, true, false, false>);
  }
  return (int)hipGetLastError();
}

template <int DH>
int launch_dh(const acattn_problem& p, const acattn_fwd_out& o, bool full, hipStream_t stream) {
  const int nT = (p.L + 15) / 16;
  if (nT <= 4) return launch_nt<DH, 4>(p, o, full, stream);
  if (nT <= 8) return launch_nt<DH, 8>(p, o, full, stream);
  return launch_nt<DH, 13>(p, o, full, stream);
}

}  // namespace

int acattn_launch_fwd_fast(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream);
int acattn_launch_fwd_dma(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream);
int acattn_launch_fwd_stream(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream);

namespace {
int g_fwd_kernel = ACATTN_FWD_AUTO;
}
int acattn_fwd_kernel_choice(int which) {
  const int old = g_fwd_kernel;
  if (which >= ACATTN_FWD_AUTO && which <= ACATTN_FWD_GENERAL) g_fwd_kernel = which;
  return old;
}

int acattn_launch_fwd(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  // training hot paths (structured mask, counter RNG, gate, two_level), -100 = not applicable:
  //   streaming kernel (L <= 208; one wave per query block, operands straight from L2): the default, fastest at
  //   every measured shape (B = 512: L = 50 20.8 us against 21.8 staged and 33.3 general; L = 200, H = 64: 179 us
  //   against 536 general);
  //   LDS-staged kernels (L <= 64; LDS-DMA staging for 48 < L, register staging below): ACATTN_FWD_STAGED
  const int which = g_fwd_kernel;
  if (which != ACATTN_FWD_GENERAL) {
    if (which == ACATTN_FWD_AUTO || which == ACATTN_FWD_STREAM) {
      const int rc_stream = acattn_launch_fwd_stream(p, o, stream);
      if (rc_stream != -100) return rc_stream;
    }
    const int rc_dma = acattn_launch_fwd_dma(p, o, stream);
    if (rc_dma != -100) return rc_dma;
    const int rc_fast = acattn_launch_fwd_fast(p, o, stream);
    if (rc_fast != -100) return rc_fast;
  }
  const bool full = p.adversarial && (!p.two_level || o.after_spatial || o.before_spatial || o.perturbed_attention ||
                                      o.calibrated_attention);
  switch (p.H / p.n_heads) {
    case 16: return launch_dh<16>(p, o, full, stream);
    case 32: return launch_dh<32>(p, o, full, stream);
    case 64: return launch_dh<64>(p, o, full, stream);
    case 128: return launch_dh<128>(p, o, full, stream);
  }
  return -1;
}

int acattn_launch_rng(int B, int nh, int L, uint64_t seed, float p_drop, float* noise, uint8_t* keep_after,
                      uint8_t* keep_mask, uint8_t* keep_before, hipStream_t stream) {
  hipLaunchKernelGGL(acattn_rng_kernel, dim3(1024), dim3(256), 0, stream, B, nh, L, seed, p_drop, noise, keep_after,
                     keep_mask, keep_before);
  return (int)hipGetLastError();
}
