// Weight chunks staged through LDS once per workgroup: shared by the hidden-128 / 256 projections (acattn_proj.hip) and
// layer tail (acattn_tail.hip).  See the comment above proj_staged_fwd_kernel for the measurement behind it.
#pragma once
#include "acattn_common.h"

namespace {

// weights travel in chunks of KT = 8 contraction tiles (128 input features: 8 float4 per lane); H = 256 has two per tile
constexpr int KT = 8;

// a tile's four bias values per lane, requested BEFORE the tile's weights: the memory counter retires in order, so a
// bias asked for after the next tile's weights would make the wait for it a wait for those weights too
__device__ __forceinline__ f4 bias_tile(const float* bias, int n_out, int nt, int g) {
  f4 b = {0.f, 0.f, 0.f, 0.f};
  if (bias) {
    const int j0 = 16 * nt + 4 * g;
    if (j0 + 3 < n_out) {
      b = *(const f4u*)(bias + j0);
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) b[r] = bias[min(j0 + r, n_out - 1)];
    }
  }
  return b;
}

#ifndef ACATTN_PROJ_NWV
#define ACATTN_PROJ_NWV 4
#endif
constexpr int NWV = ACATTN_PROJ_NWV;   // waves per workgroup
// one chunk (KTV fragments x 64 lanes) + the tile's bias (4 float4, one per lane group)
template <int KTV>
constexpr int stage_f4() { return KTV * 64 + 4; }
constexpr int STAGE_F4 = stage_f4<KT>();

// KTV = fragments per chunk: 8 (128 inputs) for hidden 128 / 256, 4 for the hidden-64 layer tail
template <int KTV>
struct WeightStageT {
  static constexpr int KT = KTV, STAGE_F4 = stage_f4<KTV>();
  f4* lds;  // [2][STAGE_F4]
  int par;  // buffer the NEXT fetch reads
  int wave, lane, c, g;
  f4 pre[KT / NWV], pre_b;
  bool has_bias;

  // ask for chunk (nt, kc) of w ([n_out, ld]) -- and with kc == 0 for the tile's bias -- into registers
  __device__ __forceinline__ void request(const float* w, int ld, const float* bias, int n_out, int nt, int kc) {
    const int o = min(16 * nt + c, n_out - 1);
#pragma unroll
    for (int i = 0; i < KT / NWV; ++i) pre[i] = *(const f4*)(w + (size_t)o * ld + 16 * (KT * kc + wave + NWV * i) + 4 * g);
    has_bias = bias != nullptr && kc == 0;
    if (has_bias && wave == 0 && lane < 4) pre_b = bias_tile(bias, n_out, nt, lane);
  }
  // the transposed use: fragment t = rows 16 t + c of w, columns 16 kt + 4 g .. (a [16 KT, ld] matrix, one column block)
  __device__ __forceinline__ void request_cols(const float* w, int ld, int kt) {
#pragma unroll
    for (int i = 0; i < KT / NWV; ++i) pre[i] = *(const f4*)(w + (size_t)(16 * (wave + NWV * i) + c) * ld + 16 * kt + 4 * g);
    has_bias = false;
  }
  // park what was asked for in the buffer the fetch after next reads; then the workgroup meets
  __device__ __forceinline__ void commit() {
    f4* dst = lds + (par ^ 1) * STAGE_F4;
#pragma unroll
    for (int i = 0; i < KT / NWV; ++i) dst[(wave + NWV * i) * 64 + lane] = pre[i];
    if (has_bias && wave == 0 && lane < 4) dst[KT * 64 + lane] = pre_b;
    __syncthreads();
    par ^= 1;
  }
  __device__ __forceinline__ void fetch(f4 (&frag)[KT], f4& b) const {
    const f4* src = lds + par * STAGE_F4;
#pragma unroll
    for (int t = 0; t < KT; ++t) frag[t] = src[t * 64 + lane];
    b = src[KT * 64 + g];
  }
};

using WeightStage = WeightStageT<8>;

}  // namespace
