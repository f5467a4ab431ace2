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
  f4 wo[DT], wd[DT];
#pragma unroll
  for (int nt = 0; nt < DT; ++nt) {
    const int f0 = woff + (16 * nt) % dh + 4 * g;
    wo[nt] = *(const f4*)(P.w_order + f0);
    wd[nt] = *(const f4*)(P.w_dist + f0);
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int row = W.row[nb], b = row / P.L, i = row - b * P.L;
    float so = 0.f, sd = 0.f;
#pragma unroll
    for (int nt = 0; nt < DT; ++nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        so = fmaf(v[nb][nt][r], wo[nt][r], so);
        sd = fmaf(v[nb][nt][r], wd[nt][r], sd);
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

int rows_per_wave(int rows) {
  static const int forced = getenv("ACATTN_PROJ_ROWS_PER_WAVE") ? atoi(getenv("ACATTN_PROJ_ROWS_PER_WAVE")) : 0;  // measurements
  if (forced == 16 || forced == 32) return forced;
  // 32 rows per wave once that still gives most SIMDs a wave: each weight fragment then feeds two row blocks.  Inside
  // the training step (B = 512, L = 50: 25,600 rows, 800 waves) 1.540 against 1.581 ms per step with 16 rows per wave.
  return rows >= 16384 ? 32 : 16;
}

}  // namespace

bool acattn_proj_supported(int H, int G) { return H == 64 && G >= 0 && G <= 256; }

int acattn_launch_proj_fwd(const acattn_proj_problem& p, const acattn_proj_out& o, hipStream_t stream) {
  const int rpw = rows_per_wave(p.rows), blocks = (p.rows + rpw - 1) / rpw;
  if (rpw == 32)
    hipLaunchKernelGGL((proj_fwd_kernel<64, 2>), dim3(blocks), dim3(64), 0, stream, p, o);
  else
    hipLaunchKernelGGL((proj_fwd_kernel<64, 1>), dim3(blocks), dim3(64), 0, stream, p, o);
  return (int)hipGetLastError();
}

int acattn_launch_proj_bwd(const acattn_proj_problem& p, const acattn_proj_bwd_io& io, hipStream_t stream) {
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
