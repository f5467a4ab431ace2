// The six projections in front of the attention core as ONE launch, forward and backward (input gradients), gfx950:
//
//     mq, mk, mv = query(x), key(x), value(x)                               recbole/model/layers.py:687-689
//     qa, ka     = attack_query_transform(mq), attack_key_transform(mk)     layers.py:658-659
//     gate       = gate(mq)                                                 layers.py:887 (combine_option 'gate')
//
// As library GEMMs these are six launches of 8-9 us forward and six more per backward walk, each re-reading x / mq /
// mk from HBM.  Here one wave owns 16 (or 32) rows: x is read once, mq and mk never leave the registers between the
// product that makes them and the products that consume them.  Same register-chain scheme as acattn_tail.hip: every
// product is computed transposed (out^T = W . in^T, exact-fp32 16x16x4 MFMA) so an accumulator is directly the B
// operand of the next product; the weight fragment of an output tile is one 16-byte load per lane, the next
// product's fragments are requested before the current product's MFMAs.
//
// Backward: dmq_t = dmq + dqa . Waq + dgate . Wg,  dmk_t = dmk + dka . Wak,  dx = dmq_t . Wq + dmk_t . Wk + dmv . Wv
// with the transposed weights gathered as dwords from the row-major parameters.  dmq_t / dmk_t are written because
// they are the cotangent operands of the query / key weight gradients (acattn_linear_wgrad_grouped).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "acattn_common.h"
#include "acattn_wstage.h"

namespace {

template <int NB>
struct Rows {
  int row[NB];
  bool ok[NB];
};

template <int NB>
__device__ __forceinline__ Rows<NB> wave_rows(int R) {
  Rows<NB> w;
  const int c = threadIdx.x & 15;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int r = (blockIdx.x * NB + nb) * 16 + c;
    w.ok[nb] = r < R;
    w.row[nb] = r < R ? r : R - 1;
  }
  return w;
}

// A fragments of out^T = W . in^T for all output tiles: lane (c, g) holds W[16nt + c][16t + 4g .. +3]
template <int DT>
__device__ __forceinline__ void load_weight(const float* w, int H, int n_out, int c, int g, f4 (&frag)[DT][DT]) {
#pragma unroll
  for (int nt = 0; nt < DT; ++nt) {
    const int o = min(16 * nt + c, n_out - 1);
#pragma unroll
    for (int t = 0; t < DT; ++t) frag[nt][t] = *(const f4*)(w + (size_t)o * H + 16 * t + 4 * g);
  }
}

// A fragments of in_grad^T = W^T . out_grad^T for a square [H, H] weight: lane (c, g) holds W[16t + 4g + r][16nt + c],
// r = 0..3 (dword gathers, unconditional: a range check per element becomes a branch per load)
template <int DT>
__device__ __forceinline__ void load_weight_t(const float* w, int c, int g, f4 (&frag)[DT][DT]) {
  constexpr int H = 16 * DT;
  // contraction group t = DT - 1 is requested last: the product that starts with it has waited for the whole set
#pragma unroll
  for (int t = 0; t < DT; ++t)
#pragma unroll
    for (int nt = 0; nt < DT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) frag[nt][t][r] = w[(16 * t + 4 * g + r) * H + 16 * nt + c];
}

// contraction groups [T0, T1) of the product (all of it by default)
template <int DT, int NB, int T0 = 0, int T1 = DT>
__device__ __forceinline__ void product(const f4 (&w)[DT][DT], const f4 (&in)[NB][DT], f4 (&acc)[NB][DT]) {
#pragma unroll
  for (int t = T0; t < T1; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int nt = 0; nt < DT; ++nt)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb][nt] = mfma16(w[nt][t][r], in[nb][t][r], acc[nb][nt]);
}

template <int DT, int NB>
__device__ __forceinline__ void init_bias(const float* bias, int n_out, int g, f4 (&acc)[NB][DT]) {
#pragma unroll
  for (int nt = 0; nt < DT; ++nt) {
    f4 b;
#pragma unroll
    for (int r = 0; r < 4; ++r) b[r] = bias[min(16 * nt + 4 * g + r, n_out - 1)];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb][nt] = b;
  }
}

template <int DT, int NB>
__device__ __forceinline__ void store_rows(float* out, const Rows<NB>& W, int g, const f4 (&v)[NB][DT]) {
  constexpr int H = 16 * DT;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
    if (W.ok[nb])
#pragma unroll
      for (int t = 0; t < DT; ++t) *(f4*)(out + (size_t)W.row[nb] * H + 16 * t + 4 * g) = v[nb][t];
}

template <int DT, int NB>
__device__ __forceinline__ void load_rows(const float* in, const Rows<NB>& W, int g, f4 (&v)[NB][DT]) {
  constexpr int H = 16 * DT;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int t = 0; t < DT; ++t) v[nb][t] = *(const f4*)(in + (size_t)W.row[nb] * H + 16 * t + 4 * g);
}

// bias of a full-width product in accumulator layout: lane (c, g) holds b[16nt + 4g .. +3]
template <int DT>
__device__ __forceinline__ void load_bias(const float* bias, int g, f4 (&b)[DT]) {
#pragma unroll
  for (int nt = 0; nt < DT; ++nt) b[nt] = *(const f4*)(bias + 16 * nt + 4 * g);
}

template <int DT, int NB>
__device__ __forceinline__ void set_rows(const f4 (&b)[DT], f4 (&acc)[NB][DT]) {
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) acc[nb][nt] = b[nt];
}


// Rank-1 halves of the spatial calibrator's affines (layers.py:705-708) for the rows of this wave, from the projected
// rows still in the accumulators (v[nb][nt][r] = value[row c][feature 16 nt + 4 g + r]):
//     plane P0     [i] = -log2(e) (v_i . w_order[woff : woff + dh] + bias_o)
//     plane P0 + 1 [i] =           v_i . w_dist [woff : woff + dh] + bias_d
// into affine[b, head, plane, i] (acattn_problem.affine).  Query halves: woff = 0, biases, P0 = 0; key halves: woff =
// dh, no bias, P0 = 2.
template <int DT, int NB>
__device__ __forceinline__ void write_affine(const f4 (&v)[NB][DT], const Rows<NB>& W, const acattn_proj_problem& P, float* affine,
                                             int woff, float bias_o, float bias_d, int P0, int c, int g) {
  constexpr float kL2e = 1.44269504088896340736f;
  const int nh = P.n_heads, dh = (16 * DT) / nh, tph = dh >> 4;  // 16-feature tiles per head
  const int LP = ((P.L + 15) >> 4) << 4;
  // the two vectors' tiles: all requested up front where the registers allow it (hidden <= 128), else tile by tile
  constexpr int PRE = DT <= 8 ? DT : 1;
  f4 wo_[PRE], wd_[PRE];
  if constexpr (DT <= 8) {
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) {
      const int f0 = woff + (16 * nt) % dh + 4 * g;
      wo_[nt] = *(const f4*)(P.w_order + f0);
      wd_[nt] = *(const f4*)(P.w_dist + f0);
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int row = W.row[nb], b = row / P.L, i = row - b * P.L;
    float so = 0.f, sd = 0.f;
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) {
      f4 wo, wd;
      if constexpr (DT <= 8) {
        wo = wo_[nt];
        wd = wd_[nt];
      } else {
        const int f0 = woff + (16 * nt) % dh + 4 * g;
        wo = *(const f4*)(P.w_order + f0);
        wd = *(const f4*)(P.w_dist + f0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        so = fmaf(v[nb][nt][r], wo[r], so);
        sd = fmaf(v[nb][nt][r], wd[r], sd);
      }
      if ((nt + 1) % tph == 0) {  // head complete (wave-uniform)
        const float o = quad_sum(so), d = quad_sum(sd);
        if (g == 0 && W.ok[nb]) {
          float* pl = affine + ((size_t)(b * nh + nt / tph) * 4 + P0) * LP + i;
          pl[0] = -kL2e * (o + bias_o);
          pl[LP] = d + bias_d;
        }
        so = 0.f;
        sd = 0.f;
      }
    }
  }
}

template <int H, int NB>
__global__ void __launch_bounds__(64) proj_fwd_kernel(const acattn_proj_problem P, const acattn_proj_out O) {
  constexpr int DT = H / 16;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const Rows<NB> W = wave_rows<NB>(P.rows);
  f4 xb[NB][DT];
  f4 wa[DT][DT], wb[DT][DT];  // two fragment sets: a product's weights are requested two products ahead
  f4 ba[DT], bb[DT];          // ... and its bias with them
  f4 m[NB][DT], acc[NB][DT];

  load_rows<DT, NB>(P.x, W, g, xb);
  load_weight<DT>(P.wq, H, H, c, g, wa);
  load_bias<DT>(P.bq, g, ba);
  PIN_ORDER();
  load_weight<DT>(P.waq, H, H, c, g, wb);
  load_bias<DT>(P.baq, g, bb);
  PIN_ORDER();
  set_rows<DT, NB>(ba, m);
  product<DT, NB>(wa, xb, m);  // mq
  store_rows<DT, NB>(O.mq, W, g, m);
  if (O.affine) write_affine<DT, NB>(m, W, P, O.affine, 0, P.b_order[0], P.b_dist[0], 0, c, g);
  load_weight<DT>(P.wk, H, H, c, g, wa);
  load_bias<DT>(P.bk, g, ba);
  PIN_ORDER();
  set_rows<DT, NB>(bb, acc);
  product<DT, NB>(wb, m, acc);  // qa = attack_query_transform(mq)
  store_rows<DT, NB>(O.qa, W, g, acc);
  if (P.wg) {
    // gate logits [rows, G], G = seq_length (50 ... 200): one 16-wide output tile at a time, weight rows requested
    // two tiles ahead
    const int GT = (P.G + 15) >> 4;
    f4 gw[3][DT], gb[3];
    auto load_tile = [&](int nt, f4 (&w)[DT], f4& b) {
      nt = min(nt, GT - 1);
      const int o = min(16 * nt + c, P.G - 1);
#pragma unroll
      for (int t = 0; t < DT; ++t) w[t] = *(const f4*)(P.wg + (size_t)o * H + 16 * t + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = P.bg[min(16 * nt + 4 * g + r, P.G - 1)];
    };
    // tile nt from buffer `cur`, tile nt + 2 requested into the buffer tile nt - 1 has left (the three buffers change
    // roles from step to step: a register copy would have to wait for the load it copies)
    auto gate_step = [&](int nt, const f4 (&cw)[DT], const f4& cb, f4 (&fw)[DT], f4& fb) {
      load_tile(nt + 2, fw, fb);
      PIN_ORDER();
      f4 ga[NB][2];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        ga[nb][0] = cb;
        ga[nb][1] = f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) ga[nb][r & 1] = mfma16(cw[t][r], m[nb][t][r], ga[nb][r & 1]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        f4 v = ga[nb][0] + ga[nb][1];
        if (O.gate_prob) v = gate_value(v, 0);  // sigmoid once per (b, i, j): the heads share it (layers.py:887)
        // the lane's four logits are consecutive in the row: one (dword-aligned) 16-byte store, not four scattered ones
        const int j0 = 16 * nt + 4 * g;
        float* dst = O.gate + (size_t)W.row[nb] * P.G + j0;
        if (W.ok[nb] && j0 + 3 < P.G) {
          *(f4u*)dst = v;
        } else if (W.ok[nb]) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j0 + r < P.G) dst[r] = v[r];
        }
      }
      PIN_ORDER();
    };
    load_tile(0, gw[0], gb[0]);
    load_tile(1, gw[1], gb[1]);
    // straight-line code for the 16 possible tiles (G <= 256): around a back edge the wait-count insertion drains every
    // request at the loop header
    static_for<16>([&](auto k) {
      constexpr int K = decltype(k)::value;
      if (K < GT) gate_step(K, gw[K % 3], gb[K % 3], gw[(K + 2) % 3], gb[(K + 2) % 3]);
    });
  }
  load_weight<DT>(P.wak, H, H, c, g, wb);
  load_bias<DT>(P.bak, g, bb);
  PIN_ORDER();
  set_rows<DT, NB>(ba, m);
  product<DT, NB>(wa, xb, m);  // mk
  store_rows<DT, NB>(O.mk, W, g, m);
  if (O.affine) write_affine<DT, NB>(m, W, P, O.affine, H / P.n_heads, 0.f, 0.f, 2, c, g);
  load_weight<DT>(P.wv, H, H, c, g, wa);
  load_bias<DT>(P.bv, g, ba);
  PIN_ORDER();
  set_rows<DT, NB>(bb, acc);
  product<DT, NB>(wb, m, acc);  // ka = attack_key_transform(mk)
  store_rows<DT, NB>(O.ka, W, g, acc);
  set_rows<DT, NB>(ba, acc);
  product<DT, NB>(wa, xb, acc);  // mv
  store_rows<DT, NB>(O.mv, W, g, acc);
}

// MODE fixes which cotangents exist at compile time (a run-time test around a request makes the wait-count insertion
// assume the shorter path and drain every load in flight): 1 = all of them (the calibrated-loss walk), 2 = dqa and dka
// only (the attacked-loss walk: everything but the attack transforms is frozen), 0 = whatever IO says.
template <int H, int NB, int MODE>
__global__ void __launch_bounds__(64) proj_bwd_kernel(const acattn_proj_problem P, const acattn_proj_bwd_io IO) {
  constexpr int DT = H / 16;
  const bool h_dmq = MODE == 1 || (MODE == 0 && IO.dmq), h_dmk = MODE == 1 || (MODE == 0 && IO.dmk);
  const bool h_dmv = MODE == 1 || (MODE == 0 && IO.dmv);
  const bool h_dqa = MODE != 0 || IO.dqa, h_dka = MODE != 0 || IO.dka, h_dx = MODE != 0 || IO.dx;
  const bool h_qt = MODE != 0 || IO.dmq_total, h_kt = MODE != 0 || IO.dmk_total;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const Rows<NB> W = wave_rows<NB>(P.rows);
  // Two fragment sets.  A product's transposed weights are 64 dword gathers and s_waitcnt counts at most 63 requests,
  // so the next product's requests go out AFTER the current product's first MFMAs (whose wait then has nothing newer
  // behind it) and have the rest of that product to arrive.
  // Order of the products: Waq^T (A), gate tiles, Wq^T (B), Wak^T (A), Wk^T (B), Wv^T (A).
  f4 wa[DT][DT], wb[DT][DT];
  f4 in[NB][DT], dq[NB][DT], dk[NB][DT], dx[NB][DT];
  auto zero = [&](f4 (&v)[NB][DT]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int t = 0; t < DT; ++t) v[nb][t] = f4{0.f, 0.f, 0.f, 0.f};
  };
  if (IO.dx_init && h_dx) load_rows<DT, NB>(IO.dx_init, W, g, dx); else zero(dx);  // the residual path's share of dx
  const bool gate = MODE != 2 && IO.dgate && P.wg;
  const int GT = gate ? (P.G + 15) >> 4 : 0;
  f4 gw[2][DT];   // gate tile: A[k = 16nt + c][j = 16 kt + 4g + r] = Wg[j][k]
  f4 gd[2][NB];   //            B[j][row] = dgate[row][j]
  auto load_tile = [&](int kt, f4 (&w)[DT], f4 (&d)[NB]) {
    kt = min(kt, GT - 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 16 * kt + 4 * g + r;
      const float live = j < P.G ? 1.f : 0.f;  // the clamped element is loaded and multiplied away
      const int jc = min(j, P.G - 1);
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) w[nt][r] = P.wg[(size_t)jc * H + 16 * nt + c] * live;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) d[nb][r] = IO.dgate[(size_t)W.row[nb] * P.G + jc];
    }
  };

  // ---- d mq (total) = dmq + dqa . Waq + dgate . Wg;  dx += d mq . Wq ------------------------------------------------
  if (h_dmq) load_rows<DT, NB>(IO.dmq, W, g, dq); else zero(dq);
  if (h_dqa) {
    load_rows<DT, NB>(IO.dqa, W, g, in);
    load_weight_t<DT>(P.waq, c, g, wa);
  }
  PIN_ORDER();
  if (h_dqa) product<DT, NB, DT - 1, DT>(wa, in, dq);
  PIN_ORDER();
  if (h_dx) load_weight_t<DT>(P.wq, c, g, wb);
  if (gate) load_tile(0, gw[0], gd[0]);
  PIN_ORDER();
  if (h_dqa) product<DT, NB, 0, DT - 1>(wa, in, dq);
  if (gate) {
    // d mq += dgate . Wg: the contraction runs over the G gate outputs, 16 at a time, the next tile requested ahead
    auto gate_step = [&](int kt, const f4 (&cw)[DT], const f4 (&cd)[NB], f4 (&fw)[DT], f4 (&fd)[NB]) {
      load_tile(kt + 1, fw, fd);
      PIN_ORDER();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) dq[nb][nt] = mfma16(cw[nt][r], cd[nb][r], dq[nb][nt]);
      PIN_ORDER();
    };
    static_for<16>([&](auto k) {
      constexpr int K = decltype(k)::value;
      if (K < GT) gate_step(K, gw[K % 2], gd[K % 2], gw[(K + 1) % 2], gd[(K + 1) % 2]);
    });
  }
  if (h_qt) store_rows<DT, NB>(IO.dmq_total, W, g, dq);
  PIN_ORDER();
  if (h_dx) product<DT, NB, DT - 1, DT>(wb, dq, dx);
  PIN_ORDER();

  // ---- d mk (total) = dmk + dka . Wak;  dx += d mk . Wk + d mv . Wv ---------------------------------------------------
  if (h_dmk) load_rows<DT, NB>(IO.dmk, W, g, dk); else zero(dk);
  if (h_dka) {
    load_rows<DT, NB>(IO.dka, W, g, in);
    load_weight_t<DT>(P.wak, c, g, wa);
  }
  PIN_ORDER();
  if (h_dx) product<DT, NB, 0, DT - 1>(wb, dq, dx);
  PIN_ORDER();
  if (h_dka) product<DT, NB, DT - 1, DT>(wa, in, dk);
  PIN_ORDER();
  if (h_dx) load_weight_t<DT>(P.wk, c, g, wb);
  PIN_ORDER();
  if (h_dka) product<DT, NB, 0, DT - 1>(wa, in, dk);
  if (h_kt) store_rows<DT, NB>(IO.dmk_total, W, g, dk);
  if (h_dx) {
    PIN_ORDER();
    product<DT, NB, DT - 1, DT>(wb, dk, dx);
    PIN_ORDER();
    if (h_dmv) {
      load_rows<DT, NB>(IO.dmv, W, g, in);
      load_weight_t<DT>(P.wv, c, g, wa);
    }
    PIN_ORDER();
    product<DT, NB, 0, DT - 1>(wb, dk, dx);
    if (h_dmv) product<DT, NB>(wa, in, dx);
    store_rows<DT, NB>(IO.dx, W, g, dx);
  }
}

// =====================================================================================================================
// Hidden sizes 128 and 256 (BASELINE configs[3], [4]) [round 3].  The H = 64 kernels above hold a product's whole weight
// matrix as MFMA fragments (two sets of H^2 / 64 registers); at H = 128 that is 512 registers.  Here the weights STREAM:
// one 16-row output tile of W at a time (H / 16 float4 per lane), the next tile requested while the current one is on
// the matrix cores, two waves per SIMD covering what is left of the latency.  What stays in registers is what the chain
// needs: the input rows x, and mq / mk between the product that makes them and the products that consume them.  The
// backward uses the same streamed product on TRANSPOSED copies of the weights (a 0.5 MB workspace filled by
// transpose_weights_kernel in front of it: the dword gathers of the H = 64 backward would be 256 per product here).
// ---------------------------------------------------------------------------------------------------------------------
// timing-only probe builds (tools/probe/build_proj_variant.sh; results are WRONG with any of them):
//   ACATTN_PROJ_W0       every tile reads weight tile 0 (what the weight stream itself costs)
//   ACATTN_PROJ_NOSTORE  no output stores                (what the output traffic costs)
//   ACATTN_PROJ_NB1      one row block per wave at hidden 128
__device__ __forceinline__ void load_wchunk(const float* w, int ld, int n_out, int nt, int kc, int c, int g, f4 (&frag)[KT]) {
#ifdef ACATTN_PROJ_W0
  nt = 0;
#endif
#ifdef ACATTN_PROJ_COALESCED  // timing only: what a pre-swizzled weight layout (one contiguous KB per wave load) would cost
  const int ntc = min(nt, (n_out >> 4) - 1), lane = 16 * g + c, dt = ld / 16;
#pragma unroll
  for (int t = 0; t < KT; ++t) frag[t] = *(const f4*)(w + ((size_t)(ntc * dt + KT * kc + t) * 64 + lane) * 4);
  return;
#endif
  const int o = min(16 * nt + c, n_out - 1);
#pragma unroll
  for (int t = 0; t < KT; ++t) frag[t] = *(const f4*)(w + (size_t)o * ld + 16 * (KT * kc + t) + 4 * g);
}

template <int DT, int NB>
__device__ __forceinline__ void chunk_product(const f4 (&w)[KT], int kc, const f4 (&in)[NB][DT], f4 (&acc)[NB]) {
  static_for<DT / KT>([&](auto q) {  // kc is a run-time value in product_emit; the register index must not be
    constexpr int KC = decltype(q)::value;
    if (kc != KC) return;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = mfma16(w[t][r], in[nb][KT * KC + t][r], acc[nb]);
  });
}

// acc[nb][nt] (+)= W[16 nt ..][:] . in^T for nt = 0 .. DT - 1, kept in registers; the steps (tile, chunk) are unrolled
// (register arrays want compile-time indices), the weights one step ahead in two rotating buffers.  INIT: start from
// the bias instead of from what acc holds.
template <int DT, int NB, bool INIT>
__device__ __forceinline__ void product_keep(const float* w, const float* bias, int c, int g, const f4 (&in)[NB][DT],
                                             f4 (&acc)[NB][DT]) {
  constexpr int H = 16 * DT, KC = DT / KT;
  f4 wbuf[2][KT], bb[2];
  if constexpr (INIT) bb[0] = bias_tile(bias, H, 0, g);
  load_wchunk(w, H, H, 0, 0, c, g, wbuf[0]);
  static_for<DT * KC>([&](auto k) {
    constexpr int S = decltype(k)::value, NT = S / KC, C0 = S % KC;
    if constexpr (S + 1 < DT * KC) {
      if constexpr (INIT && (S + 1) % KC == 0) bb[((S + 1) / KC) & 1] = bias_tile(bias, H, (S + 1) / KC, g);
      load_wchunk(w, H, H, (S + 1) / KC, (S + 1) % KC, c, g, wbuf[(S + 1) & 1]);
    }
    PIN_ORDER();
    f4 a[NB];
    if constexpr (INIT && C0 == 0) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) a[nb] = bb[NT & 1];
    } else {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) a[nb] = acc[nb][NT];
    }
    chunk_product<DT, NB>(wbuf[S & 1], C0, in, a);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb][NT] = a[nb];
  });
}

// the same for ceil(n_out / 16) output tiles that are only EMITTED (stored), tile index at run time; two steps per trip
// (two tiles at H = 128, the two chunks of one tile at H = 256)
template <int DT, int NB, class Emit>
__device__ __forceinline__ void product_emit(const float* w, int ld, const float* bias, int n_out, int c, int g,
                                             const f4 (&in)[NB][DT], Emit&& emit) {
  constexpr int KC = DT / KT;
  static_assert(KC == 1 || KC == 2, "hidden 128 or 256");
  const int n_tiles = (n_out + 15) >> 4;
  f4 wa[KT], wb[KT];
  f4 ba = bias_tile(bias, n_out, 0, g), bn;
  load_wchunk(w, ld, n_out, 0, 0, c, g, wa);
  if constexpr (KC == 1) {
    for (int nt = 0; nt < n_tiles; nt += 2) {
      bn = bias_tile(bias, n_out, min(nt + 1, n_tiles - 1), g);
      load_wchunk(w, ld, n_out, min(nt + 1, n_tiles - 1), 0, c, g, wb);
      PIN_ORDER();
      {
        f4 a[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) a[nb] = ba;
        chunk_product<DT, NB>(wa, 0, in, a);
        emit(nt, a);
      }
      ba = bias_tile(bias, n_out, min(nt + 2, n_tiles - 1), g);
      load_wchunk(w, ld, n_out, min(nt + 2, n_tiles - 1), 0, c, g, wa);
      PIN_ORDER();
      if (nt + 1 < n_tiles) {
        f4 a[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) a[nb] = bn;
        chunk_product<DT, NB>(wb, 0, in, a);
        emit(nt + 1, a);
      }
    }
  } else {
    for (int nt = 0; nt < n_tiles; ++nt) {
      load_wchunk(w, ld, n_out, nt, 1, c, g, wb);
      PIN_ORDER();
      f4 a[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) a[nb] = ba;
      chunk_product<DT, NB>(wa, 0, in, a);
      ba = bias_tile(bias, n_out, min(nt + 1, n_tiles - 1), g);
      load_wchunk(w, ld, n_out, min(nt + 1, n_tiles - 1), 0, c, g, wa);
      PIN_ORDER();
      chunk_product<DT, NB>(wb, 1, in, a);
      emit(nt, a);
    }
  }
}

#ifndef ACATTN_PROJ_WAVES
#define ACATTN_PROJ_WAVES 2
#endif
template <int H, int NB>
__global__ void __launch_bounds__(64, ACATTN_PROJ_WAVES) proj_wide_fwd_kernel(const acattn_proj_problem P, const acattn_proj_out O) {
  constexpr int DT = H / 16;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const Rows<NB> W = wave_rows<NB>(P.rows);
  f4 xb[NB][DT], m[NB][DT];
  load_rows<DT, NB>(P.x, W, g, xb);
  auto store_tile = [&](float* out, int ld, int n_out, int nt, const f4 (&a)[NB]) {
    const int j0 = 16 * nt + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (!W.ok[nb]) continue;
#ifdef ACATTN_PROJ_NOSTORE
      if (a[nb][0] != 12345.678f) continue;
#endif
      float* dst = out + (size_t)W.row[nb] * ld + j0;
      if (j0 + 3 < n_out) {
#ifdef ACATTN_PROJ_NT
        __builtin_nontemporal_store(a[nb], (f4u*)dst);
#else
        *(f4u*)dst = a[nb];
#endif
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + r < n_out) dst[r] = a[nb][r];
      }
    }
  };
  product_keep<DT, NB, true>(P.wq, P.bq, c, g, xb, m);  // mq
#ifdef ACATTN_PROJ_NOSTORE
  if (m[0][0][0] == 12345.678f)
#endif
  store_rows<DT, NB>(O.mq, W, g, m);
  if (O.affine) write_affine<DT, NB>(m, W, P, O.affine, 0, P.b_order[0], P.b_dist[0], 0, c, g);
  product_emit<DT, NB>(P.waq, H, P.baq, H, c, g, m, [&](int nt, const f4 (&a)[NB]) { store_tile(O.qa, H, H, nt, a); });
  if (P.wg) {
    product_emit<DT, NB>(P.wg, H, P.bg, P.G, c, g, m, [&](int nt, const f4 (&a)[NB]) {
      f4 v[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) v[nb] = O.gate_prob ? gate_value(a[nb], 0) : a[nb];  // sigmoid once for all heads
      store_tile(O.gate, P.G, P.G, nt, v);
    });
  }
  product_keep<DT, NB, true>(P.wk, P.bk, c, g, xb, m);  // mk
#ifdef ACATTN_PROJ_NOSTORE
  if (m[0][0][0] == 12345.678f)
#endif
  store_rows<DT, NB>(O.mk, W, g, m);
  if (O.affine) write_affine<DT, NB>(m, W, P, O.affine, H / P.n_heads, 0.f, 0.f, 2, c, g);
  product_emit<DT, NB>(P.wak, H, P.bak, H, c, g, m, [&](int nt, const f4 (&a)[NB]) { store_tile(O.ka, H, H, nt, a); });
  product_emit<DT, NB>(P.wv, H, P.bv, H, c, g, xb, [&](int nt, const f4 (&a)[NB]) { store_tile(O.mv, H, H, nt, a); });
}

// ---------------------------------------------------------------------------------------------------------------------
// The same products with the weights staged through LDS, once per WORKGROUP [round 3, second form].  Measured on the
// per-wave stream above (tools/gpu_pmc_proj.sh, probe builds): the matrix pipe is busy 39 % of the time, and neither the
// weight addresses (always tile 0: same time), nor their coalescing, nor the L2 matter -- a CU's vector memory path
// delivers ~10 B/clk (MI355X_MICROARCH.md, per-instruction constants) and eight waves each fetching 1 KB per 4 NB MFMAs
// ask for 16.  Here the four waves of a workgroup fetch each weight chunk ONCE (2 float4 per lane instead of 8), park it
// in LDS in fragment order (conflict-free ds_write_b128 / ds_read_b128, LDS delivers 128+ B/clk) and all read it from
// there: a quarter of the vector-memory traffic, one workgroup barrier per 32 NB MFMAs.
// ---------------------------------------------------------------------------------------------------------------------
// acc[nb][nt] (+)= W . in^T, kept in registers (see product_keep); `next` asks for the first chunk of whatever follows
template <int DT, int NB, bool INIT, class Next>
__device__ __forceinline__ void staged_keep(WeightStage& st, const float* w, const float* bias, const f4 (&in)[NB][DT],
                                            f4 (&acc)[NB][DT], Next&& next) {
  constexpr int H = 16 * DT, KC = DT / KT;
  static_for<DT * KC>([&](auto k) {
    constexpr int S = decltype(k)::value, NT = S / KC, C0 = S % KC;
    f4 frag[KT], b;
    st.fetch(frag, b);
    if constexpr (S + 1 < DT * KC)
      st.request(w, H, INIT ? bias : nullptr, H, (S + 1) / KC, (S + 1) % KC);
    else
      next();
    PIN_ORDER();
    f4 a[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) a[nb] = (INIT && C0 == 0) ? b : acc[nb][NT];
    chunk_product<DT, NB>(frag, C0, in, a);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb][NT] = a[nb];
    PIN_ORDER();
    st.commit();
  });
}

// the same for ceil(n_out / 16) tiles that are only emitted (see product_emit)
template <int DT, int NB, class Emit, class Next>
__device__ __forceinline__ void staged_emit(WeightStage& st, const float* w, int ld, const float* bias, int n_out,
                                            const f4 (&in)[NB][DT], Emit&& emit, Next&& next) {
  constexpr int KC = DT / KT;
  const int n_tiles = (n_out + 15) >> 4;
  for (int nt = 0; nt < n_tiles; ++nt) {
    f4 a[NB];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      f4 frag[KT], b;
      st.fetch(frag, b);
      if (kc + 1 < KC)
        st.request(w, ld, bias, n_out, nt, kc + 1);
      else if (nt + 1 < n_tiles)
        st.request(w, ld, bias, n_out, nt + 1, 0);
      else
        next();
      PIN_ORDER();
      if (kc == 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) a[nb] = b;
      }
      chunk_product<DT, NB>(frag, kc, in, a);
      PIN_ORDER();
      st.commit();
    }
    emit(nt, a);
  }
}

template <int NB>
__device__ __forceinline__ Rows<NB> wg_rows(int R, int wave) {
  Rows<NB> w;
  const int c = threadIdx.x & 15;
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int r = ((blockIdx.x * NWV + wave) * NB + nb) * 16 + c;
    w.ok[nb] = r < R;
    w.row[nb] = r < R ? r : R - 1;
  }
  return w;
}

template <int H, int NB>
__global__ void __launch_bounds__(64 * NWV, ACATTN_PROJ_WAVES) proj_staged_fwd_kernel(const acattn_proj_problem P, const acattn_proj_out O) {
  constexpr int DT = H / 16;
  __shared__ f4 stage_lds[2 * STAGE_F4];
  WeightStage st;
  st.lds = stage_lds;
  st.par = 0;
  st.lane = threadIdx.x & 63;
  st.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  st.c = threadIdx.x & 15;
  st.g = (threadIdx.x >> 4) & 3;
  const int c = st.c, g = st.g;
  const Rows<NB> W = wg_rows<NB>(P.rows, st.wave);  // (a wave past the last row keeps walking: the barriers need it)
  st.request(P.wq, H, P.bq, H, 0, 0);
  f4 xb[NB][DT], m[NB][DT];
  load_rows<DT, NB>(P.x, W, g, xb);
  st.par = 1;  // the first commit fills buffer 0 ...
  st.commit();  // ... and flips back to it
  auto store_tile = [&](float* out, int ld, int n_out, int nt, const f4 (&a)[NB]) {
    const int j0 = 16 * nt + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      if (!W.ok[nb]) continue;
#ifdef ACATTN_PROJ_NOSTORE
      if (a[nb][0] != 12345.678f) continue;
#endif
      float* dst = out + (size_t)W.row[nb] * ld + j0;
      if (j0 + 3 < n_out) {
        *(f4u*)dst = a[nb];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + r < n_out) dst[r] = a[nb][r];
      }
    }
  };
  auto masked_store_rows = [&](float* out, const f4 (&v)[NB][DT]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#ifdef ACATTN_PROJ_NOSTORE
      if (W.ok[nb] && v[nb][0][0] == 12345.678f)
#else
      if (W.ok[nb])
#endif
#pragma unroll
        for (int t = 0; t < DT; ++t) *(f4*)(out + (size_t)W.row[nb] * H + 16 * t + 4 * g) = v[nb][t];
  };
  staged_keep<DT, NB, true>(st, P.wq, P.bq, xb, m, [&] { st.request(P.waq, H, P.baq, H, 0, 0); });  // mq
  masked_store_rows(O.mq, m);
  if (O.affine) write_affine<DT, NB>(m, W, P, O.affine, 0, P.b_order[0], P.b_dist[0], 0, c, g);
  staged_emit<DT, NB>(st, P.waq, H, P.baq, H, m, [&](int nt, const f4 (&a)[NB]) { store_tile(O.qa, H, H, nt, a); },
                      [&] { if (P.wg) st.request(P.wg, H, P.bg, P.G, 0, 0); else st.request(P.wk, H, P.bk, H, 0, 0); });
  if (P.wg) {
    staged_emit<DT, NB>(st, P.wg, H, P.bg, P.G, m, [&](int nt, const f4 (&a)[NB]) {
      f4 v[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) v[nb] = O.gate_prob ? gate_value(a[nb], 0) : a[nb];  // sigmoid once for all heads
      store_tile(O.gate, P.G, P.G, nt, v);
    }, [&] { st.request(P.wk, H, P.bk, H, 0, 0); });
  }
  staged_keep<DT, NB, true>(st, P.wk, P.bk, xb, m, [&] { st.request(P.wak, H, P.bak, H, 0, 0); });  // mk
  masked_store_rows(O.mk, m);
  if (O.affine) write_affine<DT, NB>(m, W, P, O.affine, H / P.n_heads, 0.f, 0.f, 2, c, g);
  staged_emit<DT, NB>(st, P.wak, H, P.bak, H, m, [&](int nt, const f4 (&a)[NB]) { store_tile(O.ka, H, H, nt, a); },
                      [&] { st.request(P.wv, H, P.bv, H, 0, 0); });
  staged_emit<DT, NB>(st, P.wv, H, P.bv, H, xb, [&](int nt, const f4 (&a)[NB]) { store_tile(O.mv, H, H, nt, a); },
                      [&] { st.has_bias = false; });
}

// transposed copies of the layer's weights for the backward: wt[m] = W_m^T as [n_in][ldt], ldt = n_out rounded up to 16
// (pad columns zero), matrices in the order Wq, Wk, Wv, Waq, Wak ([H, H] each) and Wg ([G, H]) behind them
struct TransposeJob {
  const float* src[6];
  int n_out[6];
};
__global__ void __launch_bounds__(256) transpose_weights_kernel(const TransposeJob J, int H, float* ws) {
  __shared__ float tile[16][17];
  const int mtx = blockIdx.z;
  const float* src = J.src[mtx];
  if (!src) return;
  const int n_out = J.n_out[mtx], ldt = (n_out + 15) & ~15;
  size_t off = 0;
  for (int k = 0; k < mtx; ++k) off += (size_t)H * ((J.n_out[k] + 15) & ~15);
  float* dst = ws + off;
  const int o0 = blockIdx.x * 16, i0 = blockIdx.y * 16;
  if (o0 >= ldt) return;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  tile[ty][tx] = (o0 + ty < n_out) ? src[(size_t)(o0 + ty) * H + i0 + tx] : 0.f;
  __syncthreads();
  dst[(size_t)(i0 + ty) * ldt + o0 + tx] = tile[tx][ty];
}

// dx etc. as in proj_bwd_kernel; wt = the transposed weights above.  Three row-fragment sets fit beside the weight
// buffers, so the totals are written first and read back (by the lanes that wrote them) as the sources of dx:
// dmq_total and dmk_total are required outputs here whenever dx is asked for.
template <int H, int NB, int MODE>
__global__ void __launch_bounds__(64, 2) proj_wide_bwd_kernel(const acattn_proj_problem P, const acattn_proj_bwd_io IO,
                                                              const float* __restrict__ wt) {
  constexpr int DT = H / 16;
  const bool h_dmq = MODE == 1 || (MODE == 0 && IO.dmq), h_dmk = MODE == 1 || (MODE == 0 && IO.dmk);
  const bool h_dmv = MODE == 1 || (MODE == 0 && IO.dmv);
  const bool h_dqa = MODE != 0 || IO.dqa, h_dka = MODE != 0 || IO.dka, h_dx = MODE != 0 || IO.dx;
  const bool h_qt = MODE != 0 || IO.dmq_total, h_kt = MODE != 0 || IO.dmk_total;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const Rows<NB> W = wave_rows<NB>(P.rows);
  const size_t HH = (size_t)H * H;
  const float *wqT = wt, *wkT = wt + HH, *wvT = wt + 2 * HH, *waqT = wt + 3 * HH, *wakT = wt + 4 * HH, *wgT = wt + 5 * HH;
  const int ldg = (P.G + 15) & ~15;
  f4 in[NB][DT], tot[NB][DT];
  auto zero = [&](f4 (&v)[NB][DT]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int t = 0; t < DT; ++t) v[nb][t] = f4{0.f, 0.f, 0.f, 0.f};
  };
  // ---- d mq (total) = dmq + dqa . Waq + dgate . Wg ------------------------------------------------------------------
  if (h_dmq) load_rows<DT, NB>(IO.dmq, W, g, tot); else zero(tot);
  if (h_dqa) {
    load_rows<DT, NB>(IO.dqa, W, g, in);
    product_keep<DT, NB, false>(waqT, nullptr, c, g, in, tot);
  }
  if (MODE != 2 && IO.dgate && P.wg) {
    // the contraction runs over the G gate outputs, 16 at a time: B = dgate[row][16 kt + 4 g ..], A = WgT[16 nt + c][16 kt + 4 g ..]
    const int GT = ldg >> 4;
    for (int kt = 0; kt < GT; ++kt) {
      f4 d[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int j0 = 16 * kt + 4 * g;
        const float* src = IO.dgate + (size_t)W.row[nb] * P.G + j0;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (j0 + 3 < P.G) {
          v = *(const f4u*)src;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j0 + r < P.G) v[r] = src[r];
        }
        d[nb] = v;
      }
      f4 wg4[DT];
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) wg4[nt] = *(const f4*)(wgT + (size_t)(16 * nt + c) * ldg + 16 * kt + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) tot[nb][nt] = mfma16(wg4[nt][r], d[nb][r], tot[nb][nt]);
    }
  }
  if (h_qt) store_rows<DT, NB>(IO.dmq_total, W, g, tot);
  // ---- d mk (total) = dmk + dka . Wak ---------------------------------------------------------------------------------
  if (h_dmk) load_rows<DT, NB>(IO.dmk, W, g, tot); else zero(tot);
  if (h_dka) {
    load_rows<DT, NB>(IO.dka, W, g, in);
    product_keep<DT, NB, false>(wakT, nullptr, c, g, in, tot);
  }
  if (h_kt) store_rows<DT, NB>(IO.dmk_total, W, g, tot);
  // ---- dx = dx_init + d mk . Wk + d mq . Wq + d mv . Wv  (tot still holds d mk; d mq comes back from dmq_total) ----------
  if (h_dx) {
    f4 (&dx)[NB][DT] = in;
    if (IO.dx_init) load_rows<DT, NB>(IO.dx_init, W, g, dx); else zero(dx);
    product_keep<DT, NB, false>(wkT, nullptr, c, g, tot, dx);
    load_rows<DT, NB>(IO.dmq_total, W, g, tot);
    product_keep<DT, NB, false>(wqT, nullptr, c, g, tot, dx);
    if (h_dmv) {
      load_rows<DT, NB>(IO.dmv, W, g, tot);
      product_keep<DT, NB, false>(wvT, nullptr, c, g, tot, dx);
    }
    store_rows<DT, NB>(IO.dx, W, g, dx);
  }
}

// the backward of proj_wide_bwd_kernel with the (transposed) weights staged through LDS once per workgroup
template <int H, int NB, int MODE>
__global__ void __launch_bounds__(64 * NWV, 2) proj_staged_bwd_kernel(const acattn_proj_problem P, const acattn_proj_bwd_io IO,
                                                                      const float* __restrict__ wt) {
  constexpr int DT = H / 16;
  static_assert(DT == KT || MODE >= 0, "");
  const bool h_dmq = MODE == 1 || (MODE == 0 && IO.dmq), h_dmk = MODE == 1 || (MODE == 0 && IO.dmk);
  const bool h_dmv = MODE == 1 || (MODE == 0 && IO.dmv);
  const bool h_dqa = MODE != 0 || IO.dqa, h_dka = MODE != 0 || IO.dka, h_dx = MODE != 0 || IO.dx;
  const bool h_qt = MODE != 0 || IO.dmq_total, h_kt = MODE != 0 || IO.dmk_total;
  const bool h_gate = MODE != 2 && IO.dgate && P.wg;
  __shared__ f4 stage_lds[2 * STAGE_F4];
  WeightStage st;
  st.lds = stage_lds;
  st.lane = threadIdx.x & 63;
  st.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  st.c = threadIdx.x & 15;
  st.g = (threadIdx.x >> 4) & 3;
  const int c = st.c, g = st.g;
  const Rows<NB> W = wg_rows<NB>(P.rows, st.wave);
  const size_t HH = (size_t)H * H;
  const float *wqT = wt, *wkT = wt + HH, *wvT = wt + 2 * HH, *waqT = wt + 3 * HH, *wakT = wt + 4 * HH, *wgT = wt + 5 * HH;
  const int ldg = (P.G + 15) & ~15, GT = ldg >> 4;
  // the chain of products (all flags are uniform over the launch): each asks for the first chunk of the next one
  auto first = [&](const float* wT) { st.request(wT, H, nullptr, H, 0, 0); };
  auto after_k_total = [&] { if (h_dx) first(wkT); else st.has_bias = false; };
  auto after_q_total = [&] { if (h_dka) first(wakT); else after_k_total(); };
  auto after_qa = [&] { if (h_gate) st.request_cols(wgT, ldg, 0); else after_q_total(); };
  if (h_dqa) first(waqT); else after_qa();
  st.par = 1;
  f4 in[NB][DT], tot[NB][DT];
  auto zero = [&](f4 (&v)[NB][DT]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int t = 0; t < DT; ++t) v[nb][t] = f4{0.f, 0.f, 0.f, 0.f};
  };
  auto masked_store_rows = [&](float* out, const f4 (&v)[NB][DT]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
      if (W.ok[nb])
#pragma unroll
        for (int t = 0; t < DT; ++t) *(f4*)(out + (size_t)W.row[nb] * H + 16 * t + 4 * g) = v[nb][t];
  };
  // ---- d mq (total) = dmq + dqa . Waq + dgate . Wg ------------------------------------------------------------------
  if (h_dmq) load_rows<DT, NB>(IO.dmq, W, g, tot); else zero(tot);
  if (h_dqa) load_rows<DT, NB>(IO.dqa, W, g, in);
  st.commit();
  if (h_dqa) staged_keep<DT, NB, false>(st, waqT, nullptr, in, tot, after_qa);
  if (h_gate) {
    // the contraction runs over the G gate outputs, 16 at a time: one staged chunk = the DT row tiles of WgT for that block
    static_assert(DT == KT || DT == 2 * KT, "");
    for (int kt = 0; kt < GT; ++kt) {
      f4 d[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int j0 = 16 * kt + 4 * g;
        const float* src = IO.dgate + (size_t)W.row[nb] * P.G + j0;
        f4 v = {0.f, 0.f, 0.f, 0.f};
        if (j0 + 3 < P.G) {
          v = *(const f4u*)src;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j0 + r < P.G) v[r] = src[r];
        }
        d[nb] = v;
      }
#pragma unroll
      for (int half = 0; half < DT / KT; ++half) {  // hidden 256: the 16 row tiles of a column block are two chunks
        f4 frag[KT], b;
        st.fetch(frag, b);
        if (half + 1 < DT / KT)
          st.request_cols(wgT + (size_t)(16 * KT) * ldg * (half + 1), ldg, kt);
        else if (kt + 1 < GT)
          st.request_cols(wgT, ldg, kt + 1);
        else
          after_q_total();
        PIN_ORDER();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nt = 0; nt < KT; ++nt)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
              // (compile-time index: the half loop is unrolled)
              if (half == 0) tot[nb][nt] = mfma16(frag[nt][r], d[nb][r], tot[nb][nt]);
              else tot[nb][DT - KT + nt] = mfma16(frag[nt][r], d[nb][r], tot[nb][DT - KT + nt]);
            }
        PIN_ORDER();
        st.commit();
      }
    }
  }
  if (h_qt) masked_store_rows(IO.dmq_total, tot);
  // ---- d mk (total) = dmk + dka . Wak ---------------------------------------------------------------------------------
  if (h_dmk) load_rows<DT, NB>(IO.dmk, W, g, tot); else zero(tot);
  if (h_dka) {
    load_rows<DT, NB>(IO.dka, W, g, in);
    staged_keep<DT, NB, false>(st, wakT, nullptr, in, tot, after_k_total);
  }
  if (h_kt) masked_store_rows(IO.dmk_total, tot);
  // ---- dx = dx_init + d mk . Wk + d mq . Wq + d mv . Wv  (tot still holds d mk; d mq comes back from dmq_total) ----------
  if (h_dx) {
    f4 (&dx)[NB][DT] = in;
    if (IO.dx_init) load_rows<DT, NB>(IO.dx_init, W, g, dx); else zero(dx);
    staged_keep<DT, NB, false>(st, wkT, nullptr, tot, dx, [&] { first(wqT); });
    load_rows<DT, NB>(IO.dmq_total, W, g, tot);
    staged_keep<DT, NB, false>(st, wqT, nullptr, tot, dx, [&] { if (h_dmv) first(wvT); else st.has_bias = false; });
    if (h_dmv) {
      load_rows<DT, NB>(IO.dmv, W, g, tot);
      staged_keep<DT, NB, false>(st, wvT, nullptr, tot, dx, [&] { st.has_bias = false; });
    }
    masked_store_rows(IO.dx, dx);
  }
}

int rows_per_wave(int rows) {
  static const int forced = getenv("ACATTN_PROJ_ROWS_PER_WAVE") ? atoi(getenv("ACATTN_PROJ_ROWS_PER_WAVE")) : 0;  // measurements
  if (forced == 16 || forced == 32) return forced;
  // 32 rows per wave once that still gives most SIMDs a wave: each weight fragment then feeds two row blocks.  Inside
  // the training step (B = 512, L = 50: 25,600 rows, 800 waves) 1.540 against 1.581 ms per step with 16 rows per wave.
  return rows >= 16384 ? 32 : 16;
}

}  // namespace

bool acattn_proj_supported(int H, int G) { return (H == 64 || H == 128 || H == 256) && G >= 0 && G <= 256; }

int64_t acattn_proj_bwd_ws_bytes(const acattn_proj_problem& p) {
  if (p.H == 64) return 0;
  return ((int64_t)5 * p.H * p.H + (int64_t)p.H * ((p.G + 15) & ~15)) * (int64_t)sizeof(float);
}

namespace {
template <int H>
int launch_wide_fwd(const acattn_proj_problem& p, const acattn_proj_out& o, hipStream_t stream) {
#ifdef ACATTN_PROJ_NB1
  constexpr int NB = 1;
#else
  constexpr int NB = H == 128 ? 2 : 1;
#endif
  static const bool per_wave = getenv("ACATTN_PROJ_PER_WAVE") != nullptr;  // measurement: the per-wave weight stream
  if (!per_wave) {
    const int blocks = (p.rows + 16 * NB * NWV - 1) / (16 * NB * NWV);
    hipLaunchKernelGGL((proj_staged_fwd_kernel<H, NB>), dim3(blocks), dim3(64 * NWV), 0, stream, p, o);
    return (int)hipGetLastError();
  }
  const int blocks = (p.rows + 16 * NB - 1) / (16 * NB);
  hipLaunchKernelGGL((proj_wide_fwd_kernel<H, NB>), dim3(blocks), dim3(64), 0, stream, p, o);
  return (int)hipGetLastError();
}
template <int H>
int launch_wide_bwd(const acattn_proj_problem& p, const acattn_proj_bwd_io& io, hipStream_t stream) {
  constexpr int NB = H == 128 ? 2 : 1;
  if (!io.workspace) {
    acattn_set_error("projections backward at hidden 128 / 256 needs acattn_proj_bwd_io.workspace");
    return -1;
  }
  if (io.dx && !io.dmq_total) {
    acattn_set_error("projections backward at hidden 128 / 256: dx needs dmq_total (it is read back as a source of dx)");
    return -1;
  }
  float* wt = (float*)io.workspace;
  TransposeJob J;
  const float* src[6] = {p.wq, p.wk, p.wv, p.waq, p.wak, p.wg};
  for (int k = 0; k < 6; ++k) {
    J.src[k] = src[k];
    J.n_out[k] = k < 5 ? p.H : p.G;
  }
  const int max_out = std::max(p.H, (p.G + 15) & ~15);
  hipLaunchKernelGGL(transpose_weights_kernel, dim3(max_out / 16, p.H / 16, p.wg ? 6 : 5), dim3(256), 0, stream, J, p.H, wt);
  const bool outs = io.dx && io.dmq_total && io.dmk_total, attack = io.dqa && io.dka;
  const bool all_in = attack && io.dmq && io.dmk && io.dmv && (io.dgate || !p.wg);
  const bool attack_only = attack && !io.dmq && !io.dmk && !io.dmv && !io.dgate;
  const int mode = !outs ? 0 : all_in ? 1 : attack_only ? 2 : 0;
  static const bool per_wave = getenv("ACATTN_PROJ_PER_WAVE") != nullptr;  // measurement: the per-wave weight stream
  if (!per_wave) {
    const int wgs = (p.rows + 16 * NB * NWV - 1) / (16 * NB * NWV);
    if (mode == 1)
      hipLaunchKernelGGL((proj_staged_bwd_kernel<H, NB, 1>), dim3(wgs), dim3(64 * NWV), 0, stream, p, io, (const float*)wt);
    else if (mode == 2)
      hipLaunchKernelGGL((proj_staged_bwd_kernel<H, NB, 2>), dim3(wgs), dim3(64 * NWV), 0, stream, p, io, (const float*)wt);
    else
      hipLaunchKernelGGL((proj_staged_bwd_kernel<H, NB, 0>), dim3(wgs), dim3(64 * NWV), 0, stream, p, io, (const float*)wt);
    return (int)hipGetLastError();
  }
  const int blocks = (p.rows + 16 * NB - 1) / (16 * NB);
  if (mode == 1)
    hipLaunchKernelGGL((proj_wide_bwd_kernel<H, NB, 1>), dim3(blocks), dim3(64), 0, stream, p, io, (const float*)wt);
  else if (mode == 2)
    hipLaunchKernelGGL((proj_wide_bwd_kernel<H, NB, 2>), dim3(blocks), dim3(64), 0, stream, p, io, (const float*)wt);
  else
    hipLaunchKernelGGL((proj_wide_bwd_kernel<H, NB, 0>), dim3(blocks), dim3(64), 0, stream, p, io, (const float*)wt);
  return (int)hipGetLastError();
}
}  // namespace

int acattn_launch_proj_fwd(const acattn_proj_problem& p, const acattn_proj_out& o, hipStream_t stream) {
  if (p.H == 128) return launch_wide_fwd<128>(p, o, stream);
  if (p.H == 256) return launch_wide_fwd<256>(p, o, stream);
  const int rpw = rows_per_wave(p.rows), blocks = (p.rows + rpw - 1) / rpw;
  if (rpw == 32)
    hipLaunchKernelGGL((proj_fwd_kernel<64, 2>), dim3(blocks), dim3(64), 0, stream, p, o);
  else
    hipLaunchKernelGGL((proj_fwd_kernel<64, 1>), dim3(blocks), dim3(64), 0, stream, p, o);
  return (int)hipGetLastError();
}

int acattn_launch_proj_bwd(const acattn_proj_problem& p, const acattn_proj_bwd_io& io, hipStream_t stream) {
  if (p.H == 128) return launch_wide_bwd<128>(p, io, stream);
  if (p.H == 256) return launch_wide_bwd<256>(p, io, stream);
  const int rpw = rows_per_wave(p.rows), blocks = (p.rows + rpw - 1) / rpw;
  const bool outs = io.dx && io.dmq_total && io.dmk_total, attack = io.dqa && io.dka;
  const bool all_in = attack && io.dmq && io.dmk && io.dmv && (io.dgate || !p.wg);
  const bool attack_only = attack && !io.dmq && !io.dmk && !io.dmv && !io.dgate;
  const int mode = !outs ? 0 : all_in ? 1 : attack_only ? 2 : 0;
#define LAUNCH(NB, MODE) hipLaunchKernelGGL((proj_bwd_kernel<64, NB, MODE>), dim3(blocks), dim3(64), 0, stream, p, io)
  if (rpw == 32) {
    if (mode == 1) LAUNCH(2, 1); else if (mode == 2) LAUNCH(2, 2); else LAUNCH(2, 0);
  } else {
    if (mode == 1) LAUNCH(1, 1); else if (mode == 2) LAUNCH(1, 2); else LAUNCH(1, 0);
  }
#undef LAUNCH
  return (int)hipGetLastError();
}
