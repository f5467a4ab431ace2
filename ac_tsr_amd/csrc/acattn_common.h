// Device-side helpers shared by the forward and backward kernels of the calibrated-attention core.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts, fp32 MFMA 16x16x4, permlane swaps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "acattn.h"

typedef float f4 __attribute__((ext_vector_type(4)));
// 4 floats at dword alignment: rows of an [.., L, L] tensor with L % 4 != 0 start 8 bytes off;
// gfx950 global memory takes dword-aligned dwordx4 accesses.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

#define ACATTN_NEG_INF (-__builtin_inff())
#define ACATTN_MASK_FILL (-10000.0f)  // recbole/model/abstract_recommender.py:142
#define ACATTN_LOG_EPS (1e-24f)       // recbole/model/layers.py:719

// Lane layout used by every kernel here: lane = 16*g + c with c = lane & 15 (query row inside a
// 16-row block) and g = lane >> 4 (which quarter of every 16-key tile the lane holds).
// A query row's keys live in the registers of its four lanes {c, c+16, c+32, c+48}; the helpers
// below reduce over those four lanes with two VALU permlane swaps (no LDS traffic).
__device__ __forceinline__ float quad_sum(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

__device__ __forceinline__ float quad_max(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto s = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(s[0]), __uint_as_float(s[1]));
}

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
  // D[16x16] += A[16x4] . B[4x16]; lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
  // D register r of lane l is D[4*(l>>4) + r][l&15].  Exact fp32 (a k-ordered fmaf chain).
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Single-instruction transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp) without the denormal
// range fix-ups of __expf/__logf: every argument here is a softmax exponent, a probability or a sum of
// probabilities, far from the fp32 range limits; exp of a -10000-masked score flushes to 0 either way.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// ---------------------------------------------------------------------------------------------
// Counter-based randomness (ACATTN_RNG_COUNTER).  One call yields, for the 4 consecutive keys
// j0..j0+3 of one query row, 4 standard normals (Box-Muller) and the two dropout keep decisions
// per key.  The stream depends only on (seed, row id, key group), so the backward kernel and
// acattn_rng_materialize() regenerate it exactly.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

struct RngGroup {
  float n[4];
  uint32_t keep_after;  // bit r set = keep
  uint32_t keep_mask;
  uint32_t keep_before;
  f4 scale_after;  // keep ? keep_scale : 0   (what dropout multiplies by)
  f4 scale_mask;
};

// row_id = (b * n_heads + h) * L + i ; grp = j0 / 4.
// One multiplicative hash of (seed, row, group) seeds a xorshift32 stream; 32-bit integer multiplies run at
// quarter rate on the VALU, shifts and xors at full rate, so the stream costs ~6 full-rate ops per word.
__device__ __forceinline__ uint32_t xs32(uint32_t& x) {
  x ^= x << 13;
  x ^= x >> 17;
  x ^= x << 5;
  return x;
}

__device__ __forceinline__ RngGroup rng_group(uint64_t seed, uint32_t row_id, uint32_t grp, float p_drop,
                                              float keep_scale = 1.0f) {
  const uint32_t s_lo = (uint32_t)seed, s_hi = (uint32_t)(seed >> 32);
  uint32_t x = mix32((row_id * 64u + grp) ^ s_lo) + s_hi;
  x = x ? x : 0x6C078965u;  // xorshift has the fixed point 0
  uint32_t w[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) w[k] = xs32(x);
  RngGroup o;
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {  // Box-Muller, 24-bit uniforms
    const float u1 = (float)((w[2 * pair] >> 8) + 1u) * (1.0f / 16777216.0f);  // (0, 1]
    const float u2 = (float)(w[2 * pair + 1] >> 8) * (1.0f / 16777216.0f);     // [0, 1) revolutions
    const float rad = __builtin_amdgcn_sqrtf(-2.0f * fast_log(u1));
    o.n[2 * pair] = rad * __builtin_amdgcn_cosf(u2);
    o.n[2 * pair + 1] = rad * __builtin_amdgcn_sinf(u2);
  }
  const uint32_t thr = (uint32_t)(p_drop * 65536.0f);  // keep iff 16 random bits >= thr
  // 16-bit fields: keep_after from w4,w5; keep_mask from w6 and the low bytes left over by the uniforms;
  // keep_before (one-level configs only, dead code elsewhere) from w7,w8
  const uint32_t fa[4] = {w[4] & 0xFFFFu, w[4] >> 16, w[5] & 0xFFFFu, w[5] >> 16};
  const uint32_t fm[4] = {w[6] & 0xFFFFu, w[6] >> 16, (w[0] & 0xFFu) | ((w[1] & 0xFFu) << 8),
                          (w[2] & 0xFFu) | ((w[3] & 0xFFu) << 8)};
  const uint32_t fb[4] = {w[7] & 0xFFFFu, w[7] >> 16, w[8] & 0xFFFFu, w[8] >> 16};
  o.keep_after = 0;
  o.keep_mask = 0;
  o.keep_before = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    o.keep_after |= (fa[r] >= thr ? 1u : 0u) << r;
    o.keep_mask |= (fm[r] >= thr ? 1u : 0u) << r;
    o.keep_before |= (fb[r] >= thr ? 1u : 0u) << r;
    o.scale_after[r] = fa[r] >= thr ? keep_scale : 0.f;
    o.scale_mask[r] = fm[r] >= thr ? keep_scale : 0.f;
  }
  return o;
}

// Block index -> (b, head).  Workgroups b and b+8 share an XCD (round-robin dispatch), so the heads
// of one sequence are placed 8 blocks apart: they read the same gate-logit tile from one L2.
// Pure speed hint; any placement gives the same results.
__device__ __forceinline__ void decode_block(int bid, int B, int nh, int& b, int& h) {
  if ((B & 7) == 0) {
    const int grp = bid / (8 * nh), rem = bid - grp * 8 * nh;
    h = rem >> 3;
    b = grp * 8 + (rem & 7);
  } else {
    b = bid / nh;
    h = bid - b * nh;
  }
}

// Host-side launchers (defined in the .hip files, called from acattn_api.hip).
int acattn_launch_fwd(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream);
int acattn_launch_bwd(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_rng(int B, int nh, int L, uint64_t seed, float p_drop, float* noise, uint8_t* keep_after,
                      uint8_t* keep_mask, uint8_t* keep_before, hipStream_t stream);
int64_t acattn_ce_ws_bytes(const acattn_ce_problem& p);
int acattn_launch_ce_fwd(const acattn_ce_problem& p, void* ws, float* lse, float* row_loss, hipStream_t stream);
int acattn_launch_ce_fwd_dir(const acattn_ce_problem& p, void* ws, float* lse, float* row_loss, float* dir,
                             hipStream_t stream);
int acattn_launch_ce_bwd(const acattn_ce_problem& p, const float* lse, const float* coef, void* ws, float* d_out,
                         float* d_table, hipStream_t stream);
int acattn_launch_ln_fwd(const acattn_ln_problem& p, float* y, float* stats, hipStream_t stream);
int acattn_launch_ln_bwd(const acattn_ln_problem& p, const float* dy, const float* stats, float* dz, float* dres,
                         float* dgb_part, hipStream_t stream);
int acattn_launch_sum_rows(const float* x, float* out, int batch, int R, int C, hipStream_t stream);
void acattn_set_error(const char* msg);
int64_t acattn_linear_wgrad_ws_bytes(int64_t M, int K, int N);
int acattn_launch_linear_wgrad(const float* const* x, const float* const* dy, const int* K, const int* N,
                               float* const* dw, float* const* db, int n_items, int64_t M, void* ws, hipStream_t stream);
int acattn_launch_embed_fwd(const acattn_embed_problem& p, float* y, float* stats, hipStream_t stream);
int acattn_launch_embed_bwd(const acattn_embed_problem& p, const float* dy, const float* stats, int64_t padding_idx,
                            float* d_table, float* d_pos_part, float* dgb_part, hipStream_t stream);
int acattn_launch_penalty_fwd(const float* m, int64_t n, float* ws, float* norm, hipStream_t stream);
int acattn_launch_penalty_bwd(const float* m, const float* norm, const float* d_norm, int64_t n, float* d_m,
                              hipStream_t stream);
