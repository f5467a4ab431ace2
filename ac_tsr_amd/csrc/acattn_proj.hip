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

// A fragments of in_grad^T = W^T . out_grad^T: lane (c, g) holds W[16t + 4g + r][16nt + c], r = 0..3 (dword gathers)
template <int DT>
__device__ __forceinline__ void load_weight_t(const float* w, int H, int n_out, int c, int g, f4 (&frag)[DT][DT]) {
#pragma unroll
  for (int nt = 0; nt < DT; ++nt)
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * t + 4 * g + r;
        frag[nt][t][r] = o < n_out ? w[(size_t)o * H + 16 * nt + c] : 0.f;
      }
}

template <int DT, int NB>
__device__ __forceinline__ void product(const f4 (&w)[DT][DT], const f4 (&in)[NB][DT], f4 (&acc)[NB][DT]) {
#pragma unroll
  for (int t = 0; t < DT; ++t)
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

template <int H, int NB>
__global__ void __launch_bounds__(64) proj_fwd_kernel(const acattn_proj_problem P, const acattn_proj_out O) {
  constexpr int DT = H / 16;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const Rows<NB> W = wave_rows<NB>(P.rows);
  f4 xb[NB][DT];
  load_rows<DT, NB>(P.x, W, g, xb);
  f4 wa[DT][DT], wb[DT][DT];  // the current product's fragments / the next one's
  f4 m[NB][DT], acc[NB][DT];

  load_weight<DT>(P.wq, H, H, c, g, wa);
  load_weight<DT>(P.waq, H, H, c, g, wb);
  init_bias<DT, NB>(P.bq, H, g, m);
  product<DT, NB>(wa, xb, m);  // mq
  store_rows<DT, NB>(O.mq, W, g, m);
  init_bias<DT, NB>(P.baq, H, g, acc);
  product<DT, NB>(wb, m, acc);  // qa = attack_query_transform(mq)
  store_rows<DT, NB>(O.qa, W, g, acc);
  load_weight<DT>(P.wk, H, H, c, g, wb);
  if (P.wg) {
    // gate logits [rows, G], G = seq_length (50 ... 200): one 16-wide output tile at a time, the next tile's weight
    // rows requested before this tile's MFMAs
    const int GT = (P.G + 15) >> 4;
    f4 gw[DT], gwn[DT];
    f4 gb, gbn;
    auto load_tile = [&](int nt, f4 (&w)[DT], f4& b) {
      const int o = min(16 * nt + c, P.G - 1);
#pragma unroll
      for (int t = 0; t < DT; ++t) w[t] = *(const f4*)(P.wg + (size_t)o * H + 16 * t + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = P.bg[min(16 * nt + 4 * g + r, P.G - 1)];
    };
    load_tile(0, gw, gb);
    for (int nt = 0; nt < GT; ++nt) {
      load_tile(nt + 1 < GT ? nt + 1 : nt, gwn, gbn);
      f4 ga[NB][2];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        ga[nb][0] = gb;
        ga[nb][1] = f4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int t = 0; t < DT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) ga[nb][r & 1] = mfma16(gw[t][r], m[nb][t][r], ga[nb][r & 1]);
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const f4 v = ga[nb][0] + ga[nb][1];
        if (W.ok[nb])
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 16 * nt + 4 * g + r;
            if (j < P.G) O.gate[(size_t)W.row[nb] * P.G + j] = v[r];
          }
      }
#pragma unroll
      for (int t = 0; t < DT; ++t) gw[t] = gwn[t];
      gb = gbn;
    }
  }
  load_weight<DT>(P.wak, H, H, c, g, wa);
  init_bias<DT, NB>(P.bk, H, g, m);
  product<DT, NB>(wb, xb, m);  // mk
  store_rows<DT, NB>(O.mk, W, g, m);
  load_weight<DT>(P.wv, H, H, c, g, wb);
  init_bias<DT, NB>(P.bak, H, g, acc);
  product<DT, NB>(wa, m, acc);  // ka = attack_key_transform(mk)
  store_rows<DT, NB>(O.ka, W, g, acc);
  init_bias<DT, NB>(P.bv, H, g, acc);
  product<DT, NB>(wb, xb, acc);  // mv
  store_rows<DT, NB>(O.mv, W, g, acc);
}

template <int H, int NB>
__global__ void __launch_bounds__(64) proj_bwd_kernel(const acattn_proj_problem P, const acattn_proj_bwd_io IO) {
  constexpr int DT = H / 16;
  const int c = threadIdx.x & 15, g = threadIdx.x >> 4;
  const Rows<NB> W = wave_rows<NB>(P.rows);
  f4 wa[DT][DT], wb[DT][DT];
  f4 in[NB][DT], dq[NB][DT], dx[NB][DT];
  auto zero = [&](f4 (&v)[NB][DT]) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int t = 0; t < DT; ++t) v[nb][t] = f4{0.f, 0.f, 0.f, 0.f};
  };
  zero(dx);

  // ---- d mq (total) = dmq + dqa . Waq + dgate . Wg;  dx += d mq . Wq ------------------------------------------------
  if (IO.dmq) load_rows<DT, NB>(IO.dmq, W, g, dq); else zero(dq);
  if (IO.dqa) {
    load_weight_t<DT>(P.waq, H, H, c, g, wa);
    load_rows<DT, NB>(IO.dqa, W, g, in);
    product<DT, NB>(wa, in, dq);
  }
  if (IO.dgate && P.wg) {
    // d mq += dgate . Wg: the contraction runs over the G gate outputs, 16 at a time
    const int GT = (P.G + 15) >> 4;
    f4 gw[DT], gwn[DT];   // A[k = 16nt + c][j = 16 kt + 4g + r] = Wg[j][k]
    f4 gd[NB], gdn[NB];   // B[j][row]
    auto load_tile = [&](int kt, f4 (&w)[DT], f4 (&d)[NB]) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * kt + 4 * g + r;
        const bool in_range = j < P.G;
        const int jc = in_range ? j : P.G - 1;
#pragma unroll
        for (int nt = 0; nt < DT; ++nt) w[nt][r] = in_range ? P.wg[(size_t)jc * H + 16 * nt + c] : 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) d[nb][r] = in_range ? IO.dgate[(size_t)W.row[nb] * P.G + jc] : 0.f;
      }
    };
    load_tile(0, gw, gd);
    for (int kt = 0; kt < GT; ++kt) {
      load_tile(kt + 1 < GT ? kt + 1 : kt, gwn, gdn);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int nt = 0; nt < DT; ++nt)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) dq[nb][nt] = mfma16(gw[nt][r], gd[nb][r], dq[nb][nt]);
#pragma unroll
      for (int nt = 0; nt < DT; ++nt) gw[nt] = gwn[nt];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) gd[nb] = gdn[nb];
    }
  }
  if (IO.dmq_total) store_rows<DT, NB>(IO.dmq_total, W, g, dq);
  if (IO.dx) {
    load_weight_t<DT>(P.wq, H, H, c, g, wb);
    product<DT, NB>(wb, dq, dx);
  }

  // ---- d mk (total) = dmk + dka . Wak;  dx += d mk . Wk --------------------------------------------------------------
  if (IO.dmk) load_rows<DT, NB>(IO.dmk, W, g, dq); else zero(dq);
  if (IO.dka) {
    load_weight_t<DT>(P.wak, H, H, c, g, wa);
    load_rows<DT, NB>(IO.dka, W, g, in);
    product<DT, NB>(wa, in, dq);
  }
  if (IO.dmk_total) store_rows<DT, NB>(IO.dmk_total, W, g, dq);
  if (IO.dx) {
    load_weight_t<DT>(P.wk, H, H, c, g, wb);
    product<DT, NB>(wb, dq, dx);
    if (IO.dmv) {
      load_weight_t<DT>(P.wv, H, H, c, g, wa);
      load_rows<DT, NB>(IO.dmv, W, g, in);
      product<DT, NB>(wa, in, dx);
    }
    store_rows<DT, NB>(IO.dx, W, g, dx);
  }
}

int rows_per_wave(int rows) {
  static const int forced = getenv("ACATTN_PROJ_ROWS_PER_WAVE") ? atoi(getenv("ACATTN_PROJ_ROWS_PER_WAVE")) : 0;  // measurements
  if (forced == 16 || forced == 32) return forced;
  // 16: inside the training step (B = 512, L = 50) 1.738 against 1.759 ms per step with 32 rows per wave (twice
  // the waves, half the chain each); L = 200: 5.27 against 5.13 ms the other way round
  return rows >= 65536 ? 32 : 16;
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
  if (rpw == 32)
    hipLaunchKernelGGL((proj_bwd_kernel<64, 2>), dim3(blocks), dim3(64), 0, stream, p, io);
  else
    hipLaunchKernelGGL((proj_bwd_kernel<64, 1>), dim3(blocks), dim3(64), 0, stream, p, io);
  return (int)hipGetLastError();
}
