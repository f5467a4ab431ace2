// Fused calibrated attention forward, streaming form: ONE WAVE per (sequence, head, 16-row query block), no LDS
// staging, no workgroup barrier.
//
// Why.  B * heads workgroups of the LDS-staged kernels (acattn_fwd_dma.hip) are exactly one resident generation
// on 256 CUs, so the chip runs "read everything, compute, write everything" and the two memory phases (~11 us at
// B = 512, L = 50) do not overlap the ~11 us of arithmetic (measured: no loads/stores 16.4 us, all three 22.5 us;
// tools/probe).  Here every wave is its own pipeline: it fetches the K / Ka / V fragments of ONE key tile at a time
// straight into the registers the MFMA reads them from (rows are shared by the four query blocks of a head through
// L2, HBM sees them once), so the memory stream of the chip is spread over the whole launch, and the dispatcher
// places the 4 * B * heads waves one by one, heaviest (most key tiles under the causal mask) first.
//
// Lane layout and arithmetic are those of the other forward kernels (acattn_common.h, acattn_fwd_body.inc):
// S^T = K.Q^T on fp32 MFMA 16x16x4, lane (c = lane & 15, g = lane >> 4) holds query row i0 + c and keys
// 16 t + 4 g + r; row reductions by permlane swaps; probabilities feed ctx^T = V^T.P^T directly.
// The key halves of the two spatial affines (one dot product per key) are formed from the K fragments the wave
// has in registers and moved to the lanes that need them with ds_bpermute (crossbar only, no LDS memory).
#include <stdlib.h>

#include <type_traits>

#include "acattn_common.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float and_bits(float x, int m) { return __uint_as_float(__float_as_uint(x) & (uint32_t)m); }

// NT = key tiles a row may span (L <= 16 NT): 4 for the shipped L = 50 configurations (4 waves per SIMD), 13 for
// L <= 208 (the row's two score tile sets are 104 registers: 2 waves per SIMD).
template <int DH, int NT>
constexpr int stream_waves() {
  return NT <= 4 ? (DH <= 32 ? 4 : 3) : 2;
}

template <int DH, int NT, bool ADV>
__global__ void __launch_bounds__(64, (stream_waves<DH, NT>())) acattn_fwd_stream_kernel(const acattn_problem P, const acattn_fwd_out O) {
  constexpr int KS = DH / 4;   // floats of a row fragment per lane
  constexpr int DT = DH / 16;  // 16-column output tiles of the context
  constexpr int NG = (NT + 3) / 4;  // 64-key groups

  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;
  const int n_items = P.B * nh;
  const bool causal = P.causal != 0;
  // heaviest query blocks first: under the causal mask block qb visits qb + 1 key tiles.  (Measured alternatives:
  // the query blocks of one item adjacent on one XCD, for L2 reuse of K / Ka / V -- 23.5 us against 21.2 at L = 50,
  // 359 against 344 at L = 200; staggered wave starts -- the launch takes exactly the stagger longer: a wave's own
  // chain of instructions and L2 round trips, not contention, sets its duration; V fragments one tile ahead in
  // pass 2 like K in pass 1 -- no change; TWO HEADS PER WAVE, one after the other, the second head's query / key
  // fragments requested under passes 2-3 of the first, mask bits and gate sigmoids shared -- 34.5 us against 22.2:
  // half as many waves with chains twice as long is the opposite of what the launch needs.)
  const int rank = blockIdx.x / n_items, item = blockIdx.x - rank * n_items;
  const int qb = causal ? nT - 1 - rank : rank;
  int b, h;
  decode_block(item, P.B, nh, b, h);
  const int lane = threadIdx.x;
  const int c = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * L;
  const int hoff = h * DH;
  const size_t bh = (size_t)b * nh + h;
  const int i0 = qb * 16, i = i0 + c;
  const bool row_ok = i < L;

  // ---- key flags, query fragments, query halves of the affines ---------------------------------------------------------
  uint8_t kvb[NG];
#pragma unroll
  for (int q = 0; q < NG; ++q) kvb[q] = 64 * q + lane < L ? P.key_valid[rowbase + 64 * q + lane] : (uint8_t)0;
  float qf[KS], qaf[KS];
  {
    const size_t off = (rowbase + (row_ok ? i : 0)) * H + hoff + KS * g;
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      const f4 t = *(const f4*)(P.q + off + 4 * s4);
      f4 ta = t;
      if (ADV) ta = *(const f4*)(P.qa + off + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf[4 * s4 + e] = row_ok ? t[e] : 0.f;
        qaf[4 * s4 + e] = (ADV && row_ok) ? ta[e] : 0.f;
      }
    }
  }
  float wko[KS], wkd[KS];  // key halves of the affine weights, this lane's slice (pass 1 only)
  float ao = 0.f, ad = 0.f;
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    const f4 a = *(const f4*)(P.w_order + KS * g + 4 * s4), d = *(const f4*)(P.w_dist + KS * g + 4 * s4);
    const f4 ak = *(const f4*)(P.w_order + DH + KS * g + 4 * s4), dk = *(const f4*)(P.w_dist + DH + KS * g + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ao += qf[4 * s4 + e] * a[e];
      ad += qf[4 * s4 + e] * d[e];
      wko[4 * s4 + e] = ak[e] * -kLog2e;  // pre-scaled: sigmoid(o) = 1 / (1 + exp2(ao2 + co2))
      wkd[4 * s4 + e] = dk[e];
    }
  }
  ao = quad_sum(ao) + P.b_order[0];
  ad = quad_sum(ad) + P.b_dist[0];
  const float sc = P.scalar[0];

  unsigned long long vk[NG];
  bool any_valid = false;
  int first_valid = L, last_valid = -1;
#pragma unroll
  for (int q = NG - 1; q >= 0; --q) {
    vk[q] = __ballot(kvb[q] != 0);
    if (vk[q]) {
      any_valid = true;
      first_valid = 64 * q + __ffsll((long long)vk[q]) - 1;
      if (last_valid < 0) last_valid = 64 * q + 63 - __clzll((long long)vk[q]);
    }
  }
  const int nt_valid = any_valid ? (last_valid >> 4) + 1 : nT;
  const bool rows_see_a_key = causal ? first_valid <= i0 : any_valid;
  const int nt = rows_see_a_key ? min(causal ? min(nT, qb + 1) : nT, nt_valid) : nT;

  // Allowed-key bits of this lane, 4 per key tile: bit 4 t + r  <=>  key 16 t + 4 g + r is a real item and (causal)
  // not after the query.  The reference adds -10000 to masked scores (abstract_recommender.py:142), so a row WITHOUT
  // an allowed key spreads over every key < L: such a row gets the "key < L" bits.  `ab` marks keys after the query
  // (layers.py:715-719 takes log(sigmoid) there, log(1 - sigmoid) elsewhere).
  unsigned long long eb = 0ull, ab = 0ull, ib = 0ull;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (16 * t < L) {
      const uint32_t vn = (uint32_t)(vk[t >> 2] >> (16 * (t & 3) + 4 * g)) & 0xFu;
      const int d0 = i - 16 * t - 4 * g;  // key r is at or before the query  <=>  r <= d0
      const uint32_t cn = d0 >= 3 ? 0xFu : (d0 < 0 ? 0u : (2u << d0) - 1u);
      const int n_in = min(max(L - 16 * t - 4 * g, 0), 4);
      eb |= (unsigned long long)(causal ? (vn & cn) : vn) << (4 * t);
      ab |= (unsigned long long)(~cn & 0xFu) << (4 * t);
      ib |= (unsigned long long)((1u << n_in) - 1u) << (4 * t);
    }
  }
  // fully masked row (left padding): uniform over every key < L.  A row's keys are spread over its four lanes.
  const bool row_dead = quad_or((uint32_t)eb | (uint32_t)(eb >> 32)) == 0u;
  if (row_dead) eb = ib;
  const uint32_t eb_lo = (uint32_t)eb, eb_hi = (uint32_t)(eb >> 32), af_lo = (uint32_t)ab, af_hi = (uint32_t)(ab >> 32);
  // under a causal mask the keys after the query carry probability exactly 0 unless the row is dead
  const bool order_select = !causal || __ballot(row_dead) != 0ull;
  auto ebit = [&](int t, int r) -> int { return sbit(t < 8 ? eb_lo : eb_hi, (4 * t + r) & 31); };
  auto abit = [&](int t, int r) -> int { return sbit(t < 8 ? af_lo : af_hi, (4 * t + r) & 31); };

  const float hs2 = 0.5f * (sc * sc);
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const float scale2 = inv_sqrt * kLog2e;
  const float ao2 = -kLog2e * ao;
  const float nc2 = -(hs2 * scale2);
  const float kNegMask2 = ACATTN_MASK_FILL * kLog2e;
  const bool has_drop = P.p_drop > 0.f;
  const float keep_scale = has_drop ? fast_rcp(1.0f - P.p_drop) : 1.0f;
  const uint32_t prow = ((uint32_t)bh * L + (row_ok ? i : 0)) * (uint32_t)L;
  const uint32_t rng_row = (uint32_t)(bh * L + i);
  const RngKey rkey = rng_key(P.seed + (P.seed_device ? *P.seed_device : 0ull));

  auto store_seg = [&](float* base, int t, const f4 val) {
    const int j0 = 16 * t + 4 * g;
    if (!row_ok || j0 >= L) return;
    float* p = base + prow + j0;
    if (j0 + 3 < L) {
      store_out4(p, val);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j0 + r < L) store_out1(p + r, val[r]);
    }
  };
  // row fragment of X in {K, Ka} for key tile t: the A operand of S^T = X.Q^T (rows past L repeat the last row: their
  // keys are masked)
  // (row offsets are formed as min(lane part + tile part, last row) in elements: an add and a min per row instead of
  // a 32-bit multiply, which issues at a quarter of the rate -- 15-19 of them per key tile and pass before)
  const int last_row = (L - 1) * H;
  const int key_lane = c * H + hoff + KS * g;      // row c of a tile, this lane's slice of the head
  const int val_lane = 4 * g * H + hoff + c;       // row 4 g of a tile, column c of the head
  const float* const kbase = P.k + rowbase * H;
  const float* const kabase = P.ka + rowbase * H;
  const float* const vbase = P.v + rowbase * H;
  auto key_frag = [&](const float* Xb, int t, f4 (&out)[KS / 4]) {
    const float* p = Xb + min(key_lane + 16 * t * H, last_row + hoff + KS * g);
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) out[s4] = *(const f4*)(p + 4 * s4);
  };
  // V^T fragment of key tile t: vf[r][dt] = V[16 t + 4 g + r][16 dt + c]
  auto value_frag = [&](int t, float (&vf)[4][DT]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* p = vbase + min(val_lane + (16 * t + r) * H, last_row + hoff + c);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) vf[r][dt] = p[16 * dt];
    }
  };

  // ---- pass 1: scores, spatial calibrator, first-level softmaxes ----------------------------------------------------------
  f4 tS[NT], tM[NT];
  float mx = ACATTN_NEG_INF, my = ACATTN_NEG_INF;
  // the fragments of tile t + 1 are requested before tile t is worked on: a wave's chain of L2 round trips would
  // otherwise be as long as its arithmetic
  f4 kq[KS / 4], kaq[KS / 4];
  key_frag(kbase, 0, kq);
  if (ADV) key_frag(kabase, 0, kaq);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt) {
      f4 k4[KS / 4], ka4[KS / 4];
#pragma unroll
      for (int s4 = 0; s4 < KS / 4; ++s4) {
        k4[s4] = kq[s4];
        if (ADV) ka4[s4] = kaq[s4];
      }
      if (t + 1 < NT && t + 1 < nt) {
        key_frag(kbase, t + 1, kq);
        if (ADV) key_frag(kabase, t + 1, kaq);
      }
      // key halves of the affines for key 16 t + c (rank-1 form of layers.py:705-708), then to the lanes of the D layout
      float co = 0.f, cd = 0.f;
      f4 aS = {0.f, 0.f, 0.f, 0.f}, aM = aS;
#pragma unroll
      for (int s4 = 0; s4 < KS / 4; ++s4) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          co += k4[s4][e] * wko[4 * s4 + e];
          cd += k4[s4][e] * wkd[4 * s4 + e];
          aS = mfma16(k4[s4][e], qf[4 * s4 + e], aS);
          if (ADV) aM = mfma16(ka4[s4][e], qaf[4 * s4 + e], aM);
        }
      }
      co = quad_sum(co);
      cd = quad_sum(cd);
      f4 co4, cd4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int src = 4 * (4 * g + r);  // byte index of lane c' = 4 g + r (its g' = 0 copy)
        co4[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, co)));
        cd4[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, cd)));
      }
      const f4 ea = co4 + ao2;
      const int d0 = i - (16 * t + 4 * g);  // distance of key r is |d0 - r|
      f4 val, lt4, mk4;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pr = fast_rcp(1.0f + ex2(ea[r]));
        val[r] = 1.0f - pr;  // layers.py:715-719: key at or before the query -> log(1 - p)
        if (order_select) val[r] = abit(t, r) ? pr : val[r];
        const int dist = abs(d0 - r) + 1;
        lt4[r] = __builtin_amdgcn_logf((float)dist) * kLn2;  // log(|i - j| + 1)   layers.py:721-723
        mk4[r] = and_bits(kNegMask2, ~ebit(t, r));
      }
      f4 lg;
#pragma unroll
      for (int r = 0; r < 4; ++r) lg[r] = __builtin_amdgcn_logf(val[r] + ACATTN_LOG_EPS);
      const f4 df = lt4 - (cd4 + ad);  // layers.py:721-726
      f4 x = aS * scale2 + mk4;        // (S + e_o + e_d) / sqrt(dh) + mask, exp2 domain
      x = lg * inv_sqrt + x;
      x = (df * df) * nc2 + x;
      tS[t] = x;
      mx = fmaxf(mx, fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
      if (ADV) {
        const f4 y = aM * scale2 + mk4;
        tM[t] = y;
        my = fmaxf(my, fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3])));
      }
    }
  }
  mx = quad_max(mx);
  if (ADV) my = quad_max(my);
  float zx = 0.f, zy = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt) {
      const f4 dx = tS[t] - mx, dy = tM[t] - my;
      f4 e, f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        e[r] = ex2(dx[r]);
        if (ADV) f[r] = ex2(dy[r]);
      }
      tS[t] = e;
      zx += (e[0] + e[1]) + (e[2] + e[3]);
      if (ADV) {
        tM[t] = f;
        zy += (f[0] + f[1]) + (f[2] + f[3]);
      }
    }
  }
  zx = quad_sum(zx);
  zy = ADV ? quad_sum(zy) : 1.f;
  const float rzx = fast_rcp(zx) * keep_scale, rzy = fast_rcp(zy) * keep_scale;  // dropout's 1 / (1 - p) folded in

  f4 ca[DT], cc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    ca[dt] = f4{0.f, 0.f, 0.f, 0.f};
    cc[dt] = ca[dt];
  }
  float zu = 0.f, zv = 0.f, zw = 0.f;

  // ---- pass 2: dropout, M out, perturbed branch into P.V, exp of the calibrated branch --------------------------------------
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt) {
      float vf[4][DT];
      value_frag(t, vf);  // in flight under the random numbers
      f4 nz = {0.f, 0.f, 0.f, 0.f};
      uint32_t ka = 0xFu, km = 0xFu;
      if (ADV || has_drop) {
        const RngGroup rg = rng_group(rkey, rng_row, (uint32_t)(4 * t + g), P.p_drop);
        nz = rg.n;
        if (has_drop) {
          ka = rg.keep_after;
          km = rg.keep_mask;
        }
      }
      f4 p = tS[t] * rzx;  // P = dropout(softmax)   layers.py:735-736
      if (has_drop) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = keep_and(p[r], ka, r);
      }
      tS[t] = p;
      f4 eu;
      if (ADV) {
        f4 m = tM[t] * rzy;  // M   layers.py:670-672
        if (has_drop) {
#pragma unroll
          for (int r = 0; r < 4; ++r) m[r] = keep_and(m[r], km, r);
        }
        store_seg(O.attack_mask, t, m);
        const f4 au = (p * m + nz * (1.0f - m)) * kLog2e;  // layers.py:918-919
        const f4 a1 = m * (-kLog2e) + kLog2e;              // exp(1 - M)
        f4 ex1;
        int eb4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          eb4[r] = ebit(t, r);
          eu[r] = and_bits(ex2(au[r]), eb4[r]);
          ex1[r] = ex2(a1[r]);
        }
        zu += (eu[0] + eu[1]) + (eu[2] + eu[3]);
        const f4 av = (p * ex1) * kLog2e;  // layers.py:920-921
        f4 ev;
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[r] = and_bits(ex2(av[r]), eb4[r]);
        zv += (ev[0] + ev[1]) + (ev[2] + ev[3]);
        tM[t] = ev;  // M is dead from here on: keep the unnormalised A_c in its registers
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          if (ADV)
            ca[dt] = mfma16(vf[r][dt], eu[r], ca[dt]);
          else
            cc[dt] = mfma16(vf[r][dt], p[r], cc[dt]);  // spatial calibrator only: ctx = P.V
        }
      }
    }
  }

  // ---- pass 3: gate combine, final softmax, calibrated branch into P.V ----------------------------------------------------
  const uint32_t coff = ((uint32_t)rowbase + i) * H + hoff + 4 * g;
  if (ADV) {
    zu = quad_sum(zu);
    const float rzu = fast_rcp(zu);
    if (row_ok) {  // the perturbed branch is complete: its stores travel under pass 3
      float* oa = O.ctx_attacked + coff;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) store_out4(oa + 16 * dt, ca[dt] * rzu);
    }
    zv = quad_sum(zv);
    const float rzv = fast_rcp(zv);
    const float* grow = P.gate_logits + (rowbase + (row_ok ? i : 0)) * L;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t < nt) {
        float vf[4][DT];
        value_frag(t, vf);
        const int j0 = 16 * t + 4 * g;
        f4 gl = {0.f, 0.f, 0.f, 0.f};
        if (j0 + 3 < L) {
          gl = *(const f4u*)(grow + j0);
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j0 + r < L) gl[r] = grow[j0 + r];
        }
        const f4 eg = gl * (-kLog2e);
        f4 gt;
#pragma unroll
        for (int r = 0; r < 4; ++r) gt[r] = fast_rcp(1.0f + ex2(eg[r]));  // sigmoid(gate logits)   layers.py:887
        const f4 acn = tM[t] * rzv;                                      // A_c
        const f4 aw = (gt * (tS[t] - acn) + acn) * kLog2e;               // layers.py:888, 925
        f4 ew;
#pragma unroll
        for (int r = 0; r < 4; ++r) ew[r] = and_bits(ex2(aw[r]), ebit(t, r));
        zw += (ew[0] + ew[1]) + (ew[2] + ew[3]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) cc[dt] = mfma16(vf[r][dt], ew[r], cc[dt]);
      }
    }
    zw = quad_sum(zw);
    const float rzw = fast_rcp(zw);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) cc[dt] *= rzw;
    for (int t = nt; t < nT; ++t) store_seg(O.attack_mask, t, f4{0.f, 0.f, 0.f, 0.f});  // skipped tiles
  }

  if (row_ok) {
    float* oc = O.ctx_calibrated + coff;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) store_out4(oc + 16 * dt, cc[dt]);
    if (ADV && O.row_stats && g == 0) {
      // natural-log normalisers, as the backward expects them (acattn_bwd.hip): its recomputation adds the -10000
      // of a dead row to every score, so that shift goes into the stored normalisers
      float* sp = O.row_stats + (bh * L + i) * ACATTN_NSTAT;
      const float sh = row_dead ? ACATTN_MASK_FILL : 0.f;
      store_out4(sp, f4{(mx + __builtin_amdgcn_logf(zx)) * kLn2 + sh, (my + __builtin_amdgcn_logf(zy)) * kLn2 + sh,
                        sh + fast_log(zu), sh + fast_log(zv)});
      store_out4(sp + 4, f4{sh + fast_log(zw), 0.f, 0.f, 0.f});
    }
  }
}

template <int DH, int NT>
int launch_stream_nt(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  const int nT = (p.L + 15) / 16;
  const dim3 grid(p.B * p.n_heads * nT), block(64);
  if (p.adversarial)
    hipLaunchKernelGGL((acattn_fwd_stream_kernel<DH, NT, true>), grid, block, 0, stream, p, o);
  else
    hipLaunchKernelGGL((acattn_fwd_stream_kernel<DH, NT, false>), grid, block, 0, stream, p, o);
  return (int)hipGetLastError();
}

template <int DH>
int launch_stream(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  return p.L <= 64 ? launch_stream_nt<DH, 4>(p, o, stream) : launch_stream_nt<DH, 13>(p, o, stream);
}

}  // namespace

// Returns -100 when the problem is outside this kernel's domain (the caller then tries the other kernels).
int acattn_launch_fwd_stream(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  static const bool enabled = getenv("ACATTN_STREAM") ? atoi(getenv("ACATTN_STREAM")) != 0 : true;
  const bool ok = enabled && p.L <= 208 && (int64_t)p.B * p.n_heads * p.L * p.L < (1LL << 30) &&
                  (int64_t)p.B * p.L * p.H < (1LL << 30) && p.mask_mode == ACATTN_MASK_STRUCTURED &&
                  p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist &&
                  (!p.adversarial || (p.combine_option == ACATTN_COMBINE_GATE && p.two_level)) && !o.after_spatial &&
                  !o.before_spatial && !o.perturbed_attention && !o.calibrated_attention;
  if (!ok) return -100;
  switch (p.H / p.n_heads) {
    case 16: return launch_stream<16>(p, o, stream);
    case 32: return launch_stream<32>(p, o, stream);
    case 64: return launch_stream<64>(p, o, stream);
  }
  return -100;
}
