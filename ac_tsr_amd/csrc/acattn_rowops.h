// Row-wise helpers shared by the LayerNorm-shaped kernels (acattn_ln.hip, acattn_embed.hip): a row of H floats
// lives in H/4 adjacent lanes, 16 bytes per lane; reductions are DPP inside a 16-lane row plus shuffles.
#pragma once
#include "acattn_common.h"

__device__ __forceinline__ float dpp_row_sum(float v) {  // sum over the 16 lanes of a DPP row
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  return v;
}

// sum over the LPR = H/4 lanes that hold one row (LPR in {16, 32, 64}; lanes of a row are contiguous)
template <int LPR>
__device__ __forceinline__ float row_sum(float v) {
  v = dpp_row_sum(v);
  if (LPR >= 32) v += __shfl_xor(v, 16);
  if (LPR >= 64) v += __shfl_xor(v, 32);
  return v;
}

// keep-scale (1/(1-p) or 0) for the 4 consecutive columns 4*c4 .. of `row`: explicit bytes, or the counter RNG
__device__ __forceinline__ f4 row_keep_scale(float p_drop, const uint8_t* keep, uint64_t seed, int row, int c4, int H) {
  f4 k = {1.f, 1.f, 1.f, 1.f};
  if (p_drop <= 0.f) return k;
  const float ks = 1.0f / (1.0f - p_drop);
  if (keep) {
    const uint8_t* kp = keep + (size_t)row * H + 4 * c4;
#pragma unroll
    for (int e = 0; e < 4; ++e) k[e] = kp[e] ? ks : 0.f;
    return k;
  }
  const RngKey key = rng_key(seed);  // wave-uniform: scalar unit; every key bit changes with the replay step
  const uint32_t x = mix32(((uint32_t)row * (uint32_t)(H / 4) + (uint32_t)c4) ^ key.a) + key.b;
  const uint32_t w0 = x, w1 = rng_word(x, 0x68E31DA4u, 0x9E3779B1u);
  const uint32_t thr = (uint32_t)(p_drop * 65536.0f);
  k[0] = (w0 & 0xFFFFu) >= thr ? ks : 0.f;
  k[1] = (w0 >> 16) >= thr ? ks : 0.f;
  k[2] = (w1 & 0xFFFFu) >= thr ? ks : 0.f;
  k[3] = (w1 >> 16) >= thr ? ks : 0.f;
  return k;
}

