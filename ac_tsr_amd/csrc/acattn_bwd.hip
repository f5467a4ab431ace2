// Fused calibrated multi-head self-attention, backward, for gfx950 (MI355X).
//
// Same decomposition as the forward (acattn_fwd.hip): one workgroup per (sequence, head), each wave
// walks 16-row query blocks with whole rows in registers in key-major MFMA layout.  Nothing of size
// L x L is read except the incoming cotangent of M (and explicit-mode randomness): scores are
// recomputed on the matrix cores and every probability tensor is re-derived from the per-row
// log-normalisers the forward saved (row_stats), flash-attention style.
//
//   query-side gradients (dq, dqa) : dS'^T / dSa^T accumulators are directly the B operand of
//                                    dq^T = K^T . dS'^T (K, Ka staged transposable in LDS);
//   key-side gradients (dk, dka, dv): the four register tiles dS', dSa, A_p, A_comb go once through a
//                                    per-wave LDS scratch (transpose), are multiplied with Q / Qa /
//                                    dctx rows on the matrix cores and summed over query blocks
//                                    with LDS float atomics; one coalesced store at the end;
//   spatial-calibrator parameters   : rank-1 structure again -- row sums (da) and column sums (dc) of
//                                    the affine cotangents, reduced to per-(b,head) partials.
//
// Derivation (per query row, vectors over keys j; c = 1/sqrt(dh), m = additive mask):
//   x = (S + e_o + e_d) c + m      Pt = softmax(x)   P = keepA * Pt / (1-p)
//   y = Sa c + m                   Mt = softmax(y)   M = keepM * Mt / (1-p)
//   u = P M + n (1 - M) + m        A_p = softmax(u)
//   v = P exp(1 - M) + m           A_c = softmax(v)
//   A_g = combine(P, A_c)          A_w = softmax(A_g + m)
//   ctx_a = A_p V, ctx_c = A_w V   (recbole/model/layers.py:695-740, 661-672, 917-925, 677-680)
#include <stdlib.h>

#include "acattn_common.h"

namespace {

// sum over the 16 lanes of a DPP row (the 16 query rows of a block), result in every lane
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  return v;
}

// BIG = the key-side accumulators (and K/Ka) do not fit in LDS: accumulate dk/dka/dv with global float
// atomics into the (pre-zeroed) outputs and read K/Ka transposed straight from global memory.
// ONE = two_level == 0 (layers.py:911-914, 929-936): the origin attention is before_spatial (plain soft-max of the raw
// scores, its own dropout), and after_spatial enters only through the final mix
//     A_final = ratio * A_w + (1 - ratio) * after_spatial        (ratio = 0.5, or the trainable parameter).
// One more tile set (before_spatial) and one more cotangent tile set: built for L <= 64.
template <int DH, int NT, bool BIG, bool ONE = false>
__global__ void __launch_bounds__(256) acattn_bwd_kernel(const acattn_problem P, const acattn_bwd_io IO) {
  constexpr int KS = DH / 4;
  constexpr int DT = DH / 16;
  constexpr int VS = DH + 4;  // padded LDS row stride (stride % 8 == 4)

  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;
  const int LP = nT * 16;
  const int SS = 16 * (nT | 1);  // scratch row stride: 16 * odd -> transposed reads conflict-free
  int b, h;
  decode_block(blockIdx.x, P.B, nh, b, h);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * L;
  const int hoff = h * DH;
  const size_t bh = (size_t)b * nh + h;
  const bool structured = P.mask_mode == ACATTN_MASK_STRUCTURED;
  const bool use_order = P.w_order != nullptr, use_dist = P.w_dist != nullptr;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tile_floats = BIG ? 0 : LP * VS;
  float* Ks = smem;                  // [LP][VS]  K of this head      (LDS only when !BIG)
  float* Kas = Ks + tile_floats;     // [LP][VS]  Ka
  float* aK = Kas + tile_floats;     // [LP][VS]  dk accumulator
  float* aKa = aK + tile_floats;     // [LP][VS]  dka accumulator
  float* aV = aKa + tile_floats;     // [LP][VS]  dv accumulator
  float* s_co = aV + tile_floats;    // [LP]
  float* s_cd = s_co + LP;        // [LP]
  float* s_km = s_cd + LP;        // [LP]
  float* s_lt = s_km + LP;        // [LP]
  float* s_dco = s_lt + LP;       // [LP] column sums of d(order affine)
  float* s_dcd = s_dco + LP;      // [LP] column sums of d(distance affine)
  float* s_dwq = s_dcd + LP;      // [2*DH] query halves of dw_order, dw_dist
  float* s_small = s_dwq + 2 * DH;  // [8] db_order, db_dist, dscalar
  float* scratch = s_small + 8 + wave * 16 * SS;  // per wave [16][SS]

  // ---- stage K, Ka; zero the accumulators ---------------------------------------------------------
  for (int idx = threadIdx.x; !BIG && idx < LP * (DH / 4); idx += blockDim.x) {
    const int row = idx / (DH / 4), c4 = idx - row * (DH / 4);
    f4 kv = {0.f, 0.f, 0.f, 0.f}, kav = {0.f, 0.f, 0.f, 0.f};
    if (row < L) {
      kv = *(const f4*)(P.k + (rowbase + row) * H + hoff + 4 * c4);
      kav = *(const f4*)(P.ka + (rowbase + row) * H + hoff + 4 * c4);
    }
    *(f4*)(Ks + row * VS + 4 * c4) = kv;
    *(f4*)(Kas + row * VS + 4 * c4) = kav;
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    *(f4*)(aK + row * VS + 4 * c4) = z;
    *(f4*)(aKa + row * VS + 4 * c4) = z;
    *(f4*)(aV + row * VS + 4 * c4) = z;
  }
  for (int j = threadIdx.x; j < LP; j += blockDim.x) {
    float co = 0.f, cd = 0.f, km = ACATTN_NEG_INF;
    if (j < L) {
      const float* kr = P.k + (rowbase + j) * H + hoff;
      if (use_order || use_dist) {
#pragma unroll
        for (int d4 = 0; d4 < DH / 4; ++d4) {
          const f4 kv = *(const f4*)(kr + 4 * d4);
          if (use_order) {
            const f4 w = *(const f4*)(P.w_order + DH + 4 * d4);
            co += kv.x * w.x + kv.y * w.y + kv.z * w.z + kv.w * w.w;
          }
          if (use_dist) {
            const f4 w = *(const f4*)(P.w_dist + DH + 4 * d4);
            cd += kv.x * w.x + kv.y * w.y + kv.z * w.z + kv.w * w.w;
          }
        }
      }
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
    s_dco[j] = 0.f;
    s_dcd[j] = 0.f;
  }
  for (int j = threadIdx.x; j < 2 * DH + 8; j += blockDim.x) s_dwq[j] = 0.f;  // s_dwq and s_small are contiguous
  __syncthreads();

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
  const bool counter = P.rng_mode == ACATTN_RNG_COUNTER;
  const uint64_t seed_eff = P.seed + (P.seed_device ? *P.seed_device : 0ull);
  const RngKey rkey = rng_key(seed_eff);
  float acc_db_o = 0.f, acc_db_d = 0.f, acc_dsc = 0.f;  // per-lane partials, reduced at the end
  float acc_drr = 0.f;                                   // d rich_calibrated_combine_ratio (ONE)
  const float ratio = ONE ? (P.rich_combine == ACATTN_RICH_TRAINABLE ? P.rich_ratio[0] : 0.5f) : 1.0f;

  for (int kk = 0;; ++kk) {
    const int qb = (kk & 1) ? (kk + 1) * NW - 1 - wave : kk * NW + wave;
    if (qb >= nT) break;
    const int i0 = qb * 16, i = i0 + c;
    const bool row_ok = i < L;
    int nt = nT;
    if (structured && (P.causal ? first_valid <= i0 : last_valid >= 0))
      nt = min(P.causal ? min(nT, qb + 1) : nT, nt_valid);
    const size_t prow = (bh * L + (row_ok ? i : 0)) * (size_t)L;

    // ---- row fragments: q, qa and the two context cotangents, all in B-operand order ---------------
    float qf[KS], qaf[KS], gaf[KS], gcf[KS];
    {
      const size_t off = (rowbase + (row_ok ? i : 0)) * H + hoff + KS * g;
#pragma unroll
      for (int s4 = 0; s4 < KS / 4; ++s4) {
        f4 t = {0.f, 0.f, 0.f, 0.f}, ta = t, tga = t, tgc = t;
        if (row_ok) {
          t = *(const f4*)(P.q + off + 4 * s4);
          ta = *(const f4*)(P.qa + off + 4 * s4);
          if (IO.d_ctx_attacked) tga = *(const f4*)(IO.d_ctx_attacked + off + 4 * s4);
          if (IO.d_ctx_calibrated) tgc = *(const f4*)(IO.d_ctx_calibrated + off + 4 * s4);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          qf[4 * s4 + e] = t[e];
          qaf[4 * s4 + e] = ta[e];
          gaf[4 * s4 + e] = tga[e];
          gcf[4 * s4 + e] = tgc[e];
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

    f4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = {0.f, 0.f, 0.f, 0.f};
    if (row_ok) {
      const float* sp = IO.row_stats + (bh * L + i) * ACATTN_NSTAT;
      st0 = *(const f4*)sp;
      st1 = *(const f4*)(sp + 4);
    }
    const float lse_x = st0[0], lse_y = st0[1], lse_u = st0[2], lse_v = st0[3], lse_w = st1[0], lse_f = st1[1];
    const float lse_b = st1[2];  // before_spatial (written by the forward when two_level == 0)

    float mk[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
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

    // ---- four key-major products on the matrix cores: S, Sa, dA_p = V.dctx_a, dA_w = V.dctx_c --------
    f4 tS[NT], tM[NT], dAp[NT], dAw[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      tS[t] = f4{0.f, 0.f, 0.f, 0.f};
      tM[t] = tS[t];
      dAp[t] = tS[t];
      dAw[t] = tS[t];
      if (t < nt) {
        const int j = 16 * t + c;
        const bool kok = j < L;
        const size_t koff = (rowbase + (kok ? j : 0)) * H + hoff + KS * g;
        float kf[KS], kaf[KS], vf[KS];
#pragma unroll
        for (int s4 = 0; s4 < KS / 4; ++s4) {
          f4 t4 = {0.f, 0.f, 0.f, 0.f}, ta4 = t4, tv4 = t4;
          if (kok) {
            t4 = *(const f4*)(P.k + koff + 4 * s4);
            ta4 = *(const f4*)(P.ka + koff + 4 * s4);
            tv4 = *(const f4*)(P.v + koff + 4 * s4);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            kf[4 * s4 + e] = t4[e];
            kaf[4 * s4 + e] = ta4[e];
            vf[4 * s4 + e] = tv4[e];
          }
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          tS[t] = mfma16(kf[s], qf[s], tS[t]);
          tM[t] = mfma16(kaf[s], qaf[s], tM[t]);
          dAp[t] = mfma16(vf[s], gaf[s], dAp[t]);
          dAw[t] = mfma16(vf[s], gcf[s], dAw[t]);
        }
      }
    }

    // ---- randomness of this row block (same stream as the forward) ------------------------------------
    float nz[NT][4];
    uint32_t keepA = 0xFFFFFFFFu, keepM = 0xFFFFFFFFu, keepA2 = 0xFFFFFFFFu, keepM2 = 0xFFFFFFFFu;
    uint32_t keepB = 0xFFFFFFFFu;  // before_spatial's dropout (ONE: NT <= 8)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
        const int j0 = 16 * t + 4 * g;
        uint32_t ka = 0xFu, km_ = 0xFu, kb_ = 0xFu;
        if (counter) {
          const RngGroup rg = rng_group(rkey, (uint32_t)(bh * L + i), (uint32_t)(4 * t + g), P.p_drop);
#pragma unroll
          for (int r = 0; r < 4; ++r) nz[t][r] = rg.n[r];
          if (has_drop) {
            ka = rg.keep_after;
            km_ = rg.keep_mask;
            kb_ = rg.keep_before;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = j0 + r;
            const bool ok = row_ok && j < L;
            nz[t][r] = (ok && P.noise) ? P.noise[prow + j] : 0.f;
            if (has_drop && ok) {
              if (P.keep_after && !P.keep_after[prow + j]) ka &= ~(1u << r);
              if (P.keep_mask && !P.keep_mask[prow + j]) km_ &= ~(1u << r);
              if (ONE && P.keep_before && !P.keep_before[prow + j]) kb_ &= ~(1u << r);
            }
          }
        }
        if (ONE && t < 8) keepB = (keepB & ~(0xFu << (4 * t))) | (kb_ << (4 * t));
        if (t < 8) {
          keepA = (keepA & ~(0xFu << (4 * t))) | (ka << (4 * t));
          keepM = (keepM & ~(0xFu << (4 * t))) | (km_ << (4 * t));
        } else {
          keepA2 = (keepA2 & ~(0xFu << (4 * (t - 8)))) | (ka << (4 * (t - 8)));
          keepM2 = (keepM2 & ~(0xFu << (4 * (t - 8)))) | (km_ << (4 * (t - 8)));
        }
      }
    }
    auto kept = [&](uint32_t lo, uint32_t hi, int t, int r) -> bool {
      return t < 8 ? ((lo >> (4 * t + r)) & 1u) : ((hi >> (4 * (t - 8) + r)) & 1u);
    };
    auto scaleA = [&](int t, int r) -> float { return has_drop ? (kept(keepA, keepA2, t, r) ? keep_scale : 0.f) : 1.f; };
    auto scaleM = [&](int t, int r) -> float { return has_drop ? (kept(keepM, keepM2, t, r) ? keep_scale : 0.f) : 1.f; };
    auto scaleB = [&](int t, int r) -> float { return has_drop ? (((keepB >> (4 * t + r)) & 1u) ? keep_scale : 0.f) : 1.f; };
    f4 tB[ONE ? NT : 1], dAf[ONE ? NT : 1];  // before_spatial (pre-dropout), cotangent of after_spatial from the final mix
    // the ORIGIN attention of the adversarial calibrator: after_spatial (two_level) or before_spatial
    auto porg = [&](int t, int r) -> float { return ONE ? tB[ONE ? t : 0][r] * scaleB(t, r) : tS[t][r] * scaleA(t, r); };

    // ---- recompute Pt = softmax(x), Mt = softmax(y) from the saved log-normalisers ---------------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
        const f4 co4 = *(const f4*)(s_co + 16 * t + 4 * g);
        const f4 cd4 = *(const f4*)(s_cd + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * t + 4 * g + r;
          float s = tS[t][r];
          if (ONE) tB[ONE ? t : 0][r] = row_ok ? fast_exp(s * inv_sqrt + mk[t][r] - lse_b) : 0.f;
          if (use_order) {
            const float pr = fast_sigmoid(ao + co4[r]);
            const float val = (j > i) ? pr : 1.0f - pr;
            s += fast_log(val + ACATTN_LOG_EPS);
          }
          if (use_dist) {
            const int dist = i > j ? i - j : j - i;
            const float df = s_lt[dist] - (ad + cd4[r]);
            s += -0.5f * ((df * df) * s2);
          }
          const float x = s * inv_sqrt + mk[t][r];
          const float y = tM[t][r] * inv_sqrt + mk[t][r];
          tS[t][r] = row_ok ? fast_exp(x - lse_x) : 0.f;  // Pt
          tM[t][r] = row_ok ? fast_exp(y - lse_y) : 0.f;  // Mt
        }
      }
    }

    float* sc_w = scratch;
    // transposes a register tile set (query-major rows of this block) through the wave's scratch and
    // accumulates  acc[key][d] += sum_i tile[i][key] * rows[i][d]  with LDS float atomics.
    auto key_side = [&](const f4 (&tile)[NT], const float* rows, float* acc) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt) *(f4*)(sc_w + c * SS + 16 * t + 4 * g) = tile[t];
      float bv[4][DT];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int qi = i0 + 4 * s + g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) bv[s][dt] = (rows && qi < L) ? rows[(rowbase + qi) * H + hoff + 16 * dt + c] : 0.f;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < nt) {
          float a[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) a[s] = sc_w[(4 * s + g) * SS + 16 * t + c];
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            f4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 4; ++s) o = mfma16(a[s], bv[s][dt], o);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int key = 16 * t + 4 * g + r;
              if (BIG) {
                if (key < L) atomicAdd(acc + ((rowbase + key) * H + hoff + 16 * dt + c), o[r]);
              } else {
                atomicAdd(acc + key * VS + 16 * dt + c, o[r]);
              }
            }
          }
        }
      }
    };

    // ---- perturbed branch: A_p, du; first parts of dP and dM -----------------------------------------------
    f4 dPa[NT], dMa[NT];
    {
      f4 Ap[NT];
      float da = 0.f;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        Ap[t] = f4{0.f, 0.f, 0.f, 0.f};
        if (t < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = porg(t, r), m = tM[t][r] * scaleM(t, r);
            const float u = (p * m + nz[t][r] * (1.0f - m)) + mk[t][r];
            const float a = row_ok ? fast_exp(u - lse_u) : 0.f;
            Ap[t][r] = a;
            da += a * dAp[t][r];
          }
        }
      }
      da = quad_sum(da);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        dPa[t] = f4{0.f, 0.f, 0.f, 0.f};
        dMa[t] = f4{0.f, 0.f, 0.f, 0.f};
        if (t < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = porg(t, r), m = tM[t][r] * scaleM(t, r);
            const float du = Ap[t][r] * (dAp[t][r] - da);
            dPa[t][r] = du * m;
            dMa[t][r] = du * (p - nz[t][r]);
          }
        }
      }
      if (IO.d_ctx_attacked) key_side(Ap, IO.d_ctx_attacked, BIG ? IO.dv : aV);  // dv += A_p^T . dctx_a
    }

    // ---- calibrated branch: A_c, gate, A_w, dw ------------------------------------------------------------
    {
      f4 Ac[NT], Aw[NT], gt[NT];
      float dc = 0.f;
      const float rate = P.anneal_rate;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        Ac[t] = f4{0.f, 0.f, 0.f, 0.f};
        Aw[t] = Ac[t];
        gt[t] = Ac[t];
        if (t < nt) {
          f4 gl = {0.f, 0.f, 0.f, 0.f};
          if (P.combine_option == ACATTN_COMBINE_GATE) {
            const int j0 = 16 * t + 4 * g;
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
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * g + r;
            const float p = porg(t, r), m = tM[t][r] * scaleM(t, r);
            const float v_ = p * fast_exp(1.0f - m) + mk[t][r];
            const float ac = row_ok ? fast_exp(v_ - lse_v) : 0.f;
            Ac[t][r] = ac;
            float ag;
            if (P.combine_option == ACATTN_COMBINE_FIXED) {
              ag = (j < L && row_ok) ? fast_exp((p + 0.5f * ac) - lse_f) : 0.f;
              gt[t][r] = ag;  // A_g itself is what the inner softmax backward needs
            } else if (P.combine_option == ACATTN_COMBINE_GATE) {
              const float gg = fast_sigmoid(gl[r]);
              gt[t][r] = gg;
              ag = gg * p + (1.0f - gg) * ac;
            } else {
              ag = rate * p + (1.0f - rate) * ac;
            }
            const float aw = row_ok ? fast_exp((ag + mk[t][r]) - lse_w) : 0.f;
            Aw[t][r] = aw;
            if (ONE) {  // final = ratio * A_w + (1 - ratio) * after_spatial   layers.py:929-934
              const float dfin = dAw[t][r], a_after = tS[t][r] * scaleA(t, r);
              acc_drr += dfin * (aw - a_after);
              dAw[t][r] = ratio * dfin;
              dAf[ONE ? t : 0][r] = (1.0f - ratio) * dfin;
            }
            dc += aw * dAw[t][r];
          }
        }
      }
      dc = quad_sum(dc);
      // dw = A_w (dA_w - dc) = dA_g;  push it through the combine, then through softmax(v)
      float rf = 0.f;
      if (P.combine_option == ACATTN_COMBINE_FIXED) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) rf += gt[t][r] * (Aw[t][r] * (dAw[t][r] - dc));
        rf = quad_sum(rf);
      }
      float r1 = 0.f;
      f4 dAc[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        dAc[t] = f4{0.f, 0.f, 0.f, 0.f};
        if (t < nt) {
          f4 dgl = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = porg(t, r);
            const float dw = Aw[t][r] * (dAw[t][r] - dc);
            float dp, dac;
            if (P.combine_option == ACATTN_COMBINE_FIXED) {
              const float dz = gt[t][r] * (dw - rf);
              dp = dz;
              dac = 0.5f * dz;
            } else if (P.combine_option == ACATTN_COMBINE_GATE) {
              const float gg = gt[t][r];
              dgl[r] = dw * (p - Ac[t][r]) * (gg * (1.0f - gg));
              dp = gg * dw;
              dac = (1.0f - gg) * dw;
            } else {
              dp = rate * dw;
              dac = (1.0f - rate) * dw;
            }
            dPa[t][r] += dp;
            dAc[t][r] = dac;
            r1 += Ac[t][r] * dac;
          }
          if (P.combine_option == ACATTN_COMBINE_GATE && row_ok) {
            const int j0 = 16 * t + 4 * g;
            if (j0 < L) {
              float* gp = IO.dgate_logits + prow + j0;
              if (j0 + 3 < L) {
                *(f4u*)gp = dgl;
              } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  if (j0 + r < L) gp[r] = dgl[r];
              }
            }
          }
        }
      }
      if (P.combine_option == ACATTN_COMBINE_GATE && row_ok) {
        for (int t = nt; t < nT; ++t) {  // tiles skipped by the causal structure carry no gradient
          const int j0 = 16 * t + 4 * g;
          float* gp = IO.dgate_logits + prow + j0;
          for (int r = 0; r < 4; ++r)
            if (j0 + r < L) gp[r] = 0.f;
        }
      }
      r1 = quad_sum(r1);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = porg(t, r), m = tM[t][r] * scaleM(t, r);
            const float ex1 = fast_exp(1.0f - m);
            const float dv = Ac[t][r] * (dAc[t][r] - r1);
            dPa[t][r] += dv * ex1;
            dMa[t][r] -= dv * (p * ex1);
          }
        }
      }
      if (ONE) {  // the value gradient sees the FINAL attention (A_w is not read again below)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t < nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) Aw[t][r] = ratio * Aw[t][r] + (1.0f - ratio) * (tS[t][r] * scaleA(t, r));
      }
      if (IO.d_ctx_calibrated) key_side(Aw, IO.d_ctx_calibrated, BIG ? IO.dv : aV);  // dv += A_final^T . dctx_c
    }

    // ---- external cotangent of M, then back through the two first-level softmaxes ------------------------
    if (IO.d_attack_mask) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < nt) {
          const int j0 = 16 * t + 4 * g;
          if (row_ok && j0 < L) {
            const float* mp = IO.d_attack_mask + prow + j0;
            if (j0 + 3 < L) {
              dMa[t] += *(const f4u*)mp;
            } else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (j0 + r < L) dMa[t][r] += mp[r];
            }
          }
        }
      }
    }
    float r2 = 0.f, r3 = 0.f, r3b = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dMa[t][r] *= scaleM(t, r);  // d Mt
          r2 += tM[t][r] * dMa[t][r];
          if (ONE) {
            dPa[t][r] *= scaleB(t, r);                // d Bt: the origin's cotangent goes to before_spatial
            dAf[ONE ? t : 0][r] *= scaleA(t, r);      // d Pt: after_spatial only hears the final mix
            r3b += tB[ONE ? t : 0][r] * dPa[t][r];
            r3 += tS[t][r] * dAf[ONE ? t : 0][r];
          } else {
            dPa[t][r] *= scaleA(t, r);  // d Pt
            r3 += tS[t][r] * dPa[t][r];
          }
        }
      }
    }
    r2 = quad_sum(r2);
    r3 = quad_sum(r3);
    if (ONE) r3b = quad_sum(r3b);
    float da_o = 0.f, da_d = 0.f;
    f4 dco[NT], dcd[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      dco[t] = f4{0.f, 0.f, 0.f, 0.f};
      dcd[t] = dco[t];
      if (t < nt) {
        const f4 co4 = *(const f4*)(s_co + 16 * t + 4 * g);
        const f4 cd4 = *(const f4*)(s_cd + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * t + 4 * g + r;
          const float dSa = (tM[t][r] * (dMa[t][r] - r2)) * inv_sqrt;
          // dS: cotangent of the score as seen through after_spatial (the calibrator terms are additive there)
          const float dS = (tS[t][r] * ((ONE ? dAf[ONE ? t : 0][r] : dPa[t][r]) - r3)) * inv_sqrt;
          dMa[t][r] = dSa;  // from here on: dSa
          // total cotangent of the raw score (ONE: plus the path through before_spatial, which has no calibrator)
          dPa[t][r] = ONE ? dS + (tB[ONE ? t : 0][r] * (dPa[t][r] - r3b)) * inv_sqrt : dS;
          if (use_order) {
            const float pr = fast_sigmoid(ao + co4[r]);
            const float val = (j > i) ? pr : 1.0f - pr;
            const float sgn = (j > i) ? 1.0f : -1.0f;
            const float d_o = dS * (sgn * pr * (1.0f - pr)) * fast_rcp(val + ACATTN_LOG_EPS);
            dco[t][r] = d_o;
            da_o += d_o;
          }
          if (use_dist) {
            const int dist = i > j ? i - j : j - i;
            const float df = s_lt[dist] - (ad + cd4[r]);
            const float d_d = dS * (df * s2);
            dcd[t][r] = d_d;
            da_d += d_d;
            acc_dsc += dS * (-(df * df) * sc);
          }
        }
      }
    }
    da_o = quad_sum(da_o);  // d a_i  (also d bias, summed over rows below)
    da_d = quad_sum(da_d);
    if (g == 0) {
      acc_db_o += da_o;
      acc_db_d += da_d;
    }
    // column sums over the 16 query rows -> per-key accumulators; query halves of the weight gradients
    if (use_order || use_dist) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (use_order) {
              const float v = row16_sum(dco[t][r]);
              if (c == 0) atomicAdd(s_dco + 16 * t + 4 * g + r, v);
            }
            if (use_dist) {
              const float v = row16_sum(dcd[t][r]);
              if (c == 0) atomicAdd(s_dcd + 16 * t + 4 * g + r, v);
            }
          }
        }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (use_order) {
          const float v = row16_sum(da_o * qf[s]);
          if (c == 0) atomicAdd(s_dwq + KS * g + s, v);
        }
        if (use_dist) {
          const float v = row16_sum(da_d * qf[s]);
          if (c == 0) atomicAdd(s_dwq + DH + KS * g + s, v);
        }
      }
    }

    // ---- query-side gradients: dq^T = K^T . dS^T (+ rank-1 calibrator terms), dqa^T = Ka^T . dSa^T ----------
    {
      f4 oq[DT], oqa[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        oq[dt] = f4{0.f, 0.f, 0.f, 0.f};
        oqa[dt] = oq[dt];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < nt) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * t + 4 * g + r;
            if (BIG) {
              const size_t ko = (rowbase + (key < L ? key : 0)) * H + hoff + c;
#pragma unroll
              for (int dt = 0; dt < DT; ++dt) {
                const float kv = key < L ? P.k[ko + 16 * dt] : 0.f, kav = key < L ? P.ka[ko + 16 * dt] : 0.f;
                oq[dt] = mfma16(kv, dPa[t][r], oq[dt]);
                oqa[dt] = mfma16(kav, dMa[t][r], oqa[dt]);
              }
            } else {
              const float* kp = Ks + key * VS + c;
              const float* kap = Kas + key * VS + c;
#pragma unroll
              for (int dt = 0; dt < DT; ++dt) {
                oq[dt] = mfma16(kp[16 * dt], dPa[t][r], oq[dt]);
                oqa[dt] = mfma16(kap[16 * dt], dMa[t][r], oqa[dt]);
              }
            }
          }
        }
      }
      if (row_ok) {
        float* dqp = IO.dq + (rowbase + i) * H + hoff + 4 * g;
        float* dqap = IO.dqa + (rowbase + i) * H + hoff + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          f4 o = oq[dt];
          if (use_order) o += da_o * *(const f4*)(P.w_order + 16 * dt + 4 * g);
          if (use_dist) o += da_d * *(const f4*)(P.w_dist + 16 * dt + 4 * g);
          *(f4*)(dqp + 16 * dt) = o;
          *(f4*)(dqap + 16 * dt) = oqa[dt];
        }
      }
    }

    // ---- key-side gradients: dk += dS^T . q, dka += dSa^T . qa ------------------------------------------
    key_side(dPa, P.q, BIG ? IO.dk : aK);
    key_side(dMa, P.qa, BIG ? IO.dka : aKa);
  }

  // ---- reduce the per-lane scalars, then write the key-side results ------------------------------------------
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc_db_o += __shfl_xor(acc_db_o, off);
    acc_db_d += __shfl_xor(acc_db_d, off);
    acc_dsc += __shfl_xor(acc_dsc, off);
    acc_drr += __shfl_xor(acc_drr, off);
  }
  if (lane == 0) {
    atomicAdd(s_small + 0, acc_db_o);
    atomicAdd(s_small + 1, acc_db_d);
    atomicAdd(s_small + 2, acc_dsc);
    if (ONE) atomicAdd(s_small + 3, acc_drr);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < L * (DH / 4); idx += blockDim.x) {
    const int row = idx / (DH / 4), c4 = idx - row * (DH / 4);
    f4 r1k = {0.f, 0.f, 0.f, 0.f};  // rank-1 calibrator terms of dk
    if (use_order) r1k += s_dco[row] * *(const f4*)(P.w_order + DH + 4 * c4);
    if (use_dist) r1k += s_dcd[row] * *(const f4*)(P.w_dist + DH + 4 * c4);
    const size_t o = (rowbase + row) * H + hoff + 4 * c4;
    if (BIG) {
      if (use_order || use_dist) {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(IO.dk + o + e, r1k[e]);
      }
    } else {
      *(f4*)(IO.dk + o) = *(const f4*)(aK + row * VS + 4 * c4) + r1k;
      *(f4*)(IO.dka + o) = *(const f4*)(aKa + row * VS + 4 * c4);
      *(f4*)(IO.dv + o) = *(const f4*)(aV + row * VS + 4 * c4);
    }
  }
  for (int d = threadIdx.x; d < 2 * DH; d += blockDim.x) {
    // order-affine weight gradient: query half from s_dwq, key half = sum_j dc_o[j] k_j
    float wo = 0.f, wd = 0.f;
    if (d < DH) {
      wo = s_dwq[d];
      wd = s_dwq[DH + d];
    } else {
      for (int j = 0; j < L; ++j) {
        const float kv = BIG ? P.k[(rowbase + j) * H + hoff + (d - DH)] : Ks[j * VS + (d - DH)];
        wo += s_dco[j] * kv;
        wd += s_dcd[j] * kv;
      }
    }
    IO.dw_order_part[bh * (IO.part_stride ? IO.part_stride : 2 * DH) + d] = wo;
    IO.dw_dist_part[bh * (IO.part_stride ? IO.part_stride : 2 * DH) + d] = wd;
  }
  if (threadIdx.x < 4)
    IO.dsmall_part[bh * (IO.part_stride ? IO.part_stride : 4) + threadIdx.x] =
        (threadIdx.x < 3 || (ONE && P.rich_combine == ACATTN_RICH_TRAINABLE)) ? s_small[threadIdx.x] : 0.f;
}

template <int DH, int NT, bool BIG, bool ONE = false>
int launch_kernel(const acattn_problem& p, const acattn_bwd_io& io, int NW, size_t lds, hipStream_t stream) {
  auto kern = acattn_bwd_kernel<DH, NT, BIG, ONE>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(p.B * p.n_heads), dim3(64 * NW), lds, stream, p, io);
  return (int)hipGetLastError();
}

template <int DH, int NT>
int launch_nt(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  const int nT = (p.L + 15) / 16;
  const int NW = nT <= 4 ? 2 : 4;
  const int LP = nT * 16, SS = 16 * (nT | 1);
  const size_t small = (size_t)(6 * LP + 2 * DH + 8 + NW * 16 * SS) * sizeof(float);
  const size_t lds = small + (size_t)5 * LP * (DH + 4) * sizeof(float);
  if (!p.two_level) {
    if constexpr (NT == 4) return launch_kernel<DH, 4, false, true>(p, io, NW, lds, stream);
    acattn_set_error("backward with two_level = 0 supports L <= 64");
    return -1;
  }
  if (lds <= 150 * 1024) return launch_kernel<DH, NT, false>(p, io, NW, lds, stream);
  // long sequences: key-side sums go through global atomics into zeroed outputs
  const size_t bytes = (size_t)p.B * p.L * p.H * sizeof(float);
  hipError_t e = hipMemsetAsync(io.dk, 0, bytes, stream);
  if (e == hipSuccess) e = hipMemsetAsync(io.dka, 0, bytes, stream);
  if (e == hipSuccess) e = hipMemsetAsync(io.dv, 0, bytes, stream);
  if (e != hipSuccess) return (int)e;
  return launch_kernel<DH, NT, true>(p, io, NW, small, stream);
}

template <int DH>
int launch_dh(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  const int nT = (p.L + 15) / 16;
  if (nT <= 4) return launch_nt<DH, 4>(p, io, stream);
  if (nT <= 8) return launch_nt<DH, 8>(p, io, stream);
  return launch_nt<DH, 13>(p, io, stream);
}

}  // namespace

int acattn_launch_bwd_fast(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_stream(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);

namespace {
int g_bwd_kernel = ACATTN_BWD_AUTO;
}
int acattn_bwd_kernel_choice(int which) {
  const int old = g_bwd_kernel;
  if (which >= ACATTN_BWD_AUTO && which <= ACATTN_BWD_ROW) g_bwd_kernel = which;
  return old;
}

int acattn_launch_bwd(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  // training hot paths (structured mask, counter RNG, gate, two_level), -100 = not applicable:
  //   streaming two-kernel backward (acattn_bwd_stream.hip; needs io.workspace): the default.  B = 512, all three
  //   cotangents: L = 50 57 + 41 us against 131 us row-resident; L = 200 (H = 128, 4 heads) 0.87 ms against 7.0 ms;
  //   row-resident kernel, one recomputation (acattn_bwd_fast.hip, L <= 64): ACATTN_BWD_ROW, or no workspace
  const int which = g_bwd_kernel;
  if (which == ACATTN_BWD_AUTO || which == ACATTN_BWD_STREAM) {
    const int rc_stream = acattn_launch_bwd_stream(p, io, stream);
    if (rc_stream != -100) return rc_stream;
  }
  const int rc_fast = acattn_launch_bwd_fast(p, io, stream);
  if (rc_fast != -100) return rc_fast;
  switch (p.H / p.n_heads) {
    case 16: return launch_dh<16>(p, io, stream);
    case 32: return launch_dh<32>(p, io, stream);
    case 64: return launch_dh<64>(p, io, stream);
    case 128: return launch_dh<128>(p, io, stream);
  }
  return -1;
}
