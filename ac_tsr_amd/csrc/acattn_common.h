// Device-side helpers shared by the forward and backward kernels of the calibrated-attention core.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts, fp32 MFMA 16x16x4, permlane swaps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

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

__device__ __forceinline__ uint32_t quad_or(uint32_t x) {
  auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  x = r[0] | r[1];
  auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  return s[0] | s[1];
}

__device__ __forceinline__ f4 mfma16(float a, float b, f4 c) {
  // D[16x16] += A[16x4] . B[4x16]; lane l supplies A[l&15][l>>4] and B[l>>4][l&15];
  // D register r of lane l is D[4*(l>>4) + r][l&15].  Exact fp32 (a k-ordered fmaf chain).
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Requests stay where they are written.  With one wave per SIMD nothing else covers a memory latency, so weight
// fragments are requested one or two products ahead of their MFMAs; left alone, the scheduler sinks every such load
// to its first use and the wave sits in s_waitcnt vmcnt(0) in front of each product.
#define PIN_ORDER() __builtin_amdgcn_sched_barrier(0)

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): straight-line code with compile-time buffer
// rotation (a run-time loop needs register copies that wait for the loads they copy, and the wait-count insertion
// drains every request at a loop header)
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

// Single-instruction transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, ~1 ulp) without the denormal
// range fix-ups of __expf/__logf: every argument here is a softmax exponent, a probability or a sum of
// probabilities, far from the fp32 range limits; exp of a -10000-masked score flushes to 0 either way.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + fast_exp(-x)); }

// The gate of the calibrated branch, g = sigmoid(gate(mixed_query)) (layers.py:887), from what acattn_problem.gate_logits
// holds: the logits, or -- acattn_problem.gate_is_prob (wave-uniform) -- g itself, computed once by the producer.
__device__ __forceinline__ f4 gate_value(const f4 gl, int is_prob) {
  if (is_prob) return gl;
  f4 gt;
#pragma unroll
  for (int r = 0; r < 4; ++r) gt[r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(gl[r] * -1.44269504088896340736f));
  return gt;
}

// ---------------------------------------------------------------------------------------------
// Counter-based randomness (ACATTN_RNG_COUNTER).  One call yields, for the 4 consecutive keys
// j0..j0+3 of one query row, 4 standard normals (Box-Muller) and the dropout keep decisions of the
// three probability tensors.  The stream depends only on (seed, row id, key group, p_drop), so the
// backward kernel and acattn_rng_materialize() regenerate it exactly.
//
//   key   = two avalanche hashes of the 64-bit seed (wave-uniform: scalar unit).  A graph replay uses
//           seed + step, and every bit of both key words changes with it: replays draw unrelated streams.
//   x     = hash(counter ^ key.a) + key.b,   counter = row_id * 64 + group
//   words = x, hash'(x ^ c1), hash'(x ^ c2), ...   (one multiply + one shift-xor each)
//   normals: word k -> radius from its high 16 bits ((h + 0.5) / 65536, |n| <= 4.85), angle from its low 16
//   keep bits: p_drop == 0.5 (every shipped configuration of the reference) -> one random bit per decision from
//           word 2; any other p -> 16-bit fields of words 2.. compared with p * 65536.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  return x;
}

struct RngKey {
  uint32_t a, b;
};

__device__ __forceinline__ RngKey rng_key(uint64_t seed) {
  const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
  RngKey k;
  k.a = mix32(lo ^ mix32(hi ^ 0x85EBCA6Bu));
  k.b = mix32(k.a + hi + 0x9E3779B9u);
  return k;
}

// second-stage word hash: a multiply and a shift-xor on a value that is already well mixed
__device__ __forceinline__ uint32_t rng_word(uint32_t x, uint32_t salt, uint32_t mul) {
  uint32_t w = (x ^ salt) * mul;
  return w ^ (w >> 16);
}

struct RngGroup {
  f4 n;                 // 4 standard normals
  uint32_t keep_after;  // bit r set = keep (4 bits each)
  uint32_t keep_mask;
  uint32_t keep_before;
};

// row_id = (b * n_heads + h) * L + i ; grp = j0 / 4 (< 64).
__device__ __forceinline__ RngGroup rng_group(const RngKey key, uint32_t row_id, uint32_t grp, float p_drop) {
  uint32_t x = (row_id * 64u + grp) ^ key.a;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  x += key.b;
  const uint32_t w0 = x, w1 = rng_word(x, 0x68E31DA4u, 0x9E3779B1u);
  RngGroup o;
  const uint32_t wn[2] = {w0, w1};
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {  // Box-Muller
    const float u1 = fmaf((float)(wn[pair] >> 16), 1.0f / 65536.0f, 0.5f / 65536.0f);  // (0, 1)
    const float u2 = (float)(wn[pair] & 0xFFFFu) * (1.0f / 65536.0f);                  // [0, 1) revolutions
    const float rad = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u1) * (-2.0f * 0.69314718055994530942f));
    o.n[2 * pair] = rad * __builtin_amdgcn_cosf(u2);
    o.n[2 * pair + 1] = rad * __builtin_amdgcn_sinf(u2);
  }
  const uint32_t w2 = rng_word(x, 0xB5297A4Du, 0x85EBCA77u);
  if (p_drop == 0.5f) {  // wave-uniform
    o.keep_after = (w2 >> 4) & 0xFu;
    o.keep_mask = (w2 >> 12) & 0xFu;
    o.keep_before = (w2 >> 20) & 0xFu;
  } else {
    const uint32_t thr = (uint32_t)(p_drop * 65536.0f);  // keep iff 16 random bits >= thr
    const uint32_t w[6] = {w2, rng_word(x, 0x1B56C4E9u, 0xC2B2AE3Du), rng_word(x, 0x7F4A7C15u, 0x27D4EB2Fu),
                           rng_word(x, 0x94D049BBu, 0x165667B1u), rng_word(x, 0xD6E8FEB8u, 0xD3A2646Du),
                           rng_word(x, 0x3C6EF372u, 0xFD7046C5u)};
    uint32_t k[3] = {0, 0, 0};
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const uint32_t f = (r & 1) ? (w[2 * t + (r >> 1)] >> 16) : (w[2 * t + (r >> 1)] & 0xFFFFu);
        k[t] |= (f >= thr ? 1u : 0u) << r;
      }
    }
    o.keep_after = k[0];
    o.keep_mask = k[1];
    o.keep_before = k[2];
  }
  return o;
}

// p_drop == 0.5 known at compile time: the same stream as rng_group(.., 0.5f) without the general-rate code
__device__ __forceinline__ RngGroup rng_group_half(const RngKey key, uint32_t row_id, uint32_t grp) {
  uint32_t x = (row_id * 64u + grp) ^ key.a;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  x += key.b;
  const uint32_t wn[2] = {x, rng_word(x, 0x68E31DA4u, 0x9E3779B1u)};
  RngGroup o;
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {  // Box-Muller
    const float u1 = fmaf((float)(wn[pair] >> 16), 1.0f / 65536.0f, 0.5f / 65536.0f);  // (0, 1)
    const float u2 = (float)(wn[pair] & 0xFFFFu) * (1.0f / 65536.0f);                  // [0, 1) revolutions
    const float rad = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u1) * (-2.0f * 0.69314718055994530942f));
    o.n[2 * pair] = rad * __builtin_amdgcn_cosf(u2);
    o.n[2 * pair + 1] = rad * __builtin_amdgcn_sinf(u2);
  }
  const uint32_t w2 = rng_word(x, 0xB5297A4Du, 0x85EBCA77u);
  o.keep_after = (w2 >> 4) & 0xFu;
  o.keep_mask = (w2 >> 12) & 0xFu;
  o.keep_before = (w2 >> 20) & 0xFu;
  return o;
}

// The same draw as rng_group_half with the four normals as two register pairs already multiplied by `scale`
// ((cos, sin) * (radius * scale): one packed multiply per pair; the product differs from (radius * cos) * scale in
// the last bit at most).
typedef float acattn_f2 __attribute__((ext_vector_type(2)));
struct RngDraw {
  acattn_f2 n01, n23;
  uint32_t keep_after, keep_mask, keep_before;
};
__device__ __forceinline__ RngDraw rng_draw_half(const RngKey key, uint32_t row_id, uint32_t grp, float scale) {
  uint32_t x = (row_id * 64u + grp) ^ key.a;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  x += key.b;
  const uint32_t wn[2] = {x, rng_word(x, 0x68E31DA4u, 0x9E3779B1u)};
  RngDraw o;
  acattn_f2 n[2];
#pragma unroll
  for (int pair = 0; pair < 2; ++pair) {  // Box-Muller
    const float u1 = fmaf((float)(wn[pair] >> 16), 1.0f / 65536.0f, 0.5f / 65536.0f);  // (0, 1)
    const float u2 = (float)(wn[pair] & 0xFFFFu) * (1.0f / 65536.0f);                  // [0, 1) revolutions
    const float rad = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u1) * (-2.0f * 0.69314718055994530942f)) * scale;
    acattn_f2 cs = {__builtin_amdgcn_cosf(u2), __builtin_amdgcn_sinf(u2)};
    asm("" : "+v"(cs));
    n[pair] = cs * rad;
  }
  o.n01 = n[0];
  o.n23 = n[1];
  const uint32_t w2 = rng_word(x, 0xB5297A4Du, 0x85EBCA77u);
  o.keep_after = (w2 >> 4) & 0xFu;
  o.keep_mask = (w2 >> 12) & 0xFu;
  o.keep_before = (w2 >> 20) & 0xFu;
  return o;
}

// The keep decisions of the attack mask's dropout alone (keep_mask of rng_group / rng_draw_half for the same arguments):
// the streaming forward needs them before it needs the normals (M leaves early, acattn_fwd_stream.inc pass 1.5).
__device__ __forceinline__ uint32_t rng_keep_mask_half(const RngKey key, uint32_t row_id, uint32_t grp) {
  uint32_t x = (row_id * 64u + grp) ^ key.a;
  x *= 0x7feb352dU;
  x ^= x >> 15;
  x *= 0x846ca68bU;
  x ^= x >> 16;
  x += key.b;
  return (rng_word(x, 0xB5297A4Du, 0x85EBCA77u) >> 12) & 0xFu;
}

// keep bits -> what dropout multiplies by
__device__ __forceinline__ f4 keep_scale4(uint32_t bits, float keep_scale) {
  f4 s;
#pragma unroll
  for (int r = 0; r < 4; ++r) s[r] = ((bits >> r) & 1u) ? keep_scale : 0.f;
  return s;
}
// x where the keep bit is set, +0 elsewhere, without a compare/select pair: sign-extended bit field AND value
// Sign-extended bit `idx` of `bits` (0 or -1) as ONE v_bfe_i32.  The empty asm makes the value opaque: otherwise
// the optimiser rewrites "mask AND value" into and + compare + select (three instructions for two).
__device__ __forceinline__ int sbit(uint32_t bits, int idx) {
  int m = __builtin_amdgcn_sbfe(bits, idx, 1);
  asm("" : "+v"(m));
  return m;
}
__device__ __forceinline__ float keep_and(float x, uint32_t bits, int r) {
  return __uint_as_float(__float_as_uint(x) & (uint32_t)sbit(bits, r));
}

// Output stores of the streaming forward kernel: 16 bytes per lane, dword aligned, streaming ("nt") cache policy.
// The output of a forward launch (23 MB at B = 512, L = 50) is not read again by it; with the default policy it stays
// dirty in the XCD L2s until the kernel ends, where the write-back overlaps nothing.  Measured on the streaming
// kernel (tools/probe, B = 512, L = 50): default 22.7 us, nt 20.6-21.5 us, write-through (sc1) 24.7 us.  The
// LDS-staged kernels keep default-policy stores (nt cost them 1.3 us).
// (Compiler builtins, not inline assembly: the hazard padding between an MFMA and a store of its accumulator and the
// vmcnt bookkeeping stay the compiler's.)
// (the pointer is only dword aligned for rows of M with L % 4 != 0; gfx950 takes dword-aligned dwordx4 accesses, and
// the 16-byte type keeps the store one instruction)
__device__ __forceinline__ void store_out4(float* p, const f4 v) { __builtin_nontemporal_store(v, (f4*)p); }
__device__ __forceinline__ void store_out1(float* p, const float v) { __builtin_nontemporal_store(v, p); }

// Block index -> (b, head).  Workgroups b and b+8 share an XCD (round-robin dispatch), so the heads
// of one sequence are placed 8 blocks apart: they read the same gate-logit tile from one L2.
// Pure speed hint; any placement gives the same results.
// Does query block qb of sequence b carry a non-zero context cotangent?  (acattn_bwd_io.active_qblocks / read_rows)
// Does some row of query block qb of sequence b carry a non-zero CONTEXT cotangent (d_ctx_*)?
__device__ __forceinline__ bool qblock_has_ctx(const acattn_bwd_io& IO, int b, int qb) {
  if (IO.active_qblocks) return (IO.active_qblocks[b] >> qb) & 1u;
  if (IO.read_rows) {  // (a position outside [0, L) matches no block: its row is treated as carrying no cotangent)
    for (int r = 0; r < IO.n_read_rows; ++r)
      if ((int)(IO.read_rows[(size_t)b * IO.n_read_rows + r] >> 4) == qb) return true;
    return false;
  }
  return true;
}

__device__ __forceinline__ bool qblock_active(const acattn_bwd_io& IO, int b, int qb) {
  if (IO.d_attack_mask || IO.d_penalty_part) return true;  // the mask cotangent reaches every row
  if (IO.active_qblocks) return (IO.active_qblocks[b] >> qb) & 1u;
  if (IO.read_rows) {
    for (int r = 0; r < IO.n_read_rows; ++r)
      if ((int)(IO.read_rows[(size_t)b * IO.n_read_rows + r] >> 4) == qb) return true;
    return false;
  }
  return true;
}

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
int acattn_fwd_kernel_choice(int which);
int acattn_bwd_kernel_choice(int which);
int64_t acattn_bwd_stream_ws_bytes(const acattn_problem& p);
int acattn_launch_bwd(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
bool acattn_bwd_gate_summed(const acattn_problem& p, const acattn_bwd_io& io);
bool acattn_bwd_pair_supported(const acattn_problem& p, const acattn_bwd_io& io);  // acattn_bwd_io.dqa2 (acattn_bwd.hip)  // acattn_bwd_io.dgate_summed (acattn_bwd.hip)
int acattn_launch_spatial_affines(const acattn_problem& p, float* affine, hipStream_t stream);
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
int acattn_launch_sum_rows_pair(const float* x1, float* out1, int batch1, int R1, int C1, const float* x2, float* out2,
                                int batch2, int R2, int C2, hipStream_t stream);
void acattn_set_error(const char* msg);
bool acattn_proj_supported(int H, int G);
int64_t acattn_proj_bwd_ws_bytes(const acattn_proj_problem& p);
int acattn_launch_proj_fwd(const acattn_proj_problem& p, const acattn_proj_out& o, hipStream_t stream);
int acattn_launch_proj_bwd(const acattn_proj_problem& p, const acattn_proj_bwd_io& io, hipStream_t stream);
bool acattn_tail_supported(int H, int I);
int acattn_tail_bwd_partial_rows(int rows);
int acattn_tail_bwd_partial_rows_h(int rows, int H);
int64_t acattn_tail_bwd_ws_bytes(int H, int I);
int acattn_select_tail_nb(int nb);
int acattn_launch_tail_fwd(const acattn_tail_problem& p, const acattn_tail_saved& s, hipStream_t stream);
int acattn_launch_tail_bwd(const acattn_tail_problem& p, const acattn_tail_saved& s, const acattn_tail_bwd_io& io,
                           hipStream_t stream);
int64_t acattn_linear_wgrad_ws_bytes(int64_t M, int K, int N);
int acattn_launch_linear_wgrad(const float* const* x, const float* const* dy, const int* K, const int* N,
                               float* const* dw, float* const* db, int n_items, int64_t M, void* ws, hipStream_t stream,
                               int* P_out = nullptr, long long* w_off_out = nullptr, long long* b_off_out = nullptr);
int acattn_launch_linear_wgrad_reduce_many(const float* const* part_w, const float* const* part_b, const int* K, const int* N,
                                           const int* P, float* const* dw, float* const* db, int n_items,
                                           const float* const* sr_x, float* const* sr_out, const int* sr_R, const int* sr_C,
                                           int n_sr, hipStream_t stream);
int acattn_launch_embed_fwd(const acattn_embed_problem& p, float* y, float* stats, hipStream_t stream);
int acattn_launch_embed_bwd(const acattn_embed_problem& p, const float* dy, const float* stats, int64_t padding_idx,
                            float* d_table, float* d_pos_part, float* dgb_part, hipStream_t stream);
int acattn_launch_penalty_fwd(const float* m, int64_t n, float* ws, float* norm, hipStream_t stream);
int acattn_launch_penalty_partial(const float* m, int64_t n, float* part, hipStream_t stream);
// acattn_fwd_out.penalty_part: did the forward launch of this host thread fill it itself? (acattn_fwd_stream.hip)
void acattn_penalty_written_set(bool v);
bool acattn_penalty_written();
int acattn_launch_penalty_rows(const float* m, int B, int nh, int L, float* pen, hipStream_t stream);
int acattn_launch_attacked_loss_finish_rows(const float* row_loss, int B, const float* const* pen, int n_masks, int count,
                                            float weight, float* out, float* scale_buf, int n_scale, hipStream_t stream);
int acattn_launch_penalty_drows(const float* norms, const float* d_loss, float scale, int count, float* const* d_pen,
                                int n_masks, hipStream_t stream, const float* dir = nullptr, float* d_out = nullptr, int n_dir = 0);
int acattn_launch_attacked_loss_finish(const float* row_loss, int B, const float* part, int n_masks, int64_t mask_numel,
                                       float weight, float* out, float* scale_buf, int n_scale, hipStream_t stream);
int acattn_launch_penalty_bwd_scaled(const float* m, const float* norm, const float* d_loss, float scale, int64_t n,
                                     float* d_m, hipStream_t stream);
int acattn_launch_penalty_bwd(const float* m, const float* norm, const float* d_norm, int64_t n, float* d_m,
                              hipStream_t stream);
// zero fill by a kernel (acattn_util.hip: memset nodes inside a hipGraph proved unreliable for accumulate-into-zero buffers)
int acattn_launch_zero(float* p, size_t n, hipStream_t stream);
// acattn_ce_bf16.hip: the hidden-64 sweeps with split (3 x bf16) operands
int64_t acattn_ce6_rows_bytes(const acattn_ce_problem& p);
int acattn_launch_ce6_sweep(const acattn_ce_problem& p, const float* lse, const float* coef, float* slab, float* d_table,
                            float2* part, void* rows_ws, int n_wg, int n_left, bool dir, hipStream_t stream);
int acattn_launch_ce6_onehot_reduce(const acattn_ce_problem& p, const float* coef, const float* slab, int n_slabs, float* d_out,
                                    float* d_table, hipStream_t stream);
int acattn_launch_ce6_fwd_sweep(const acattn_ce_problem& p, float2* part, void* rows_ws, int n_wg, int n_left, hipStream_t stream);
int acattn_ce_products_choice(int mode);
int acattn_launch_dense_ce_fwd(const float* logits, int64_t rows, int64_t N, const int64_t* target, float* lse, float* row_loss,
                               hipStream_t stream);
int acattn_launch_dense_ce_bwd(const float* logits, const float* lse, const int64_t* target, const float* coef, int64_t rows,
                               int64_t N, float* d_logits, hipStream_t stream);
int acattn_launch_step_inputs(const void* const* src, void* const* dst, const int64_t* bytes, int n, int64_t* counter,
                              const int64_t* item_length, int64_t* last_row, int n_rows, hipStream_t stream);
int acattn_launch_penalty_partial_multi(const float* const* m, int n_masks, int64_t n, float* part, hipStream_t stream);
int acattn_launch_penalty_bwd_scaled_multi(const float* const* m, const float* norms, const float* d_loss, float scale,
                                           int64_t n, float* const* d_m, int n_masks, hipStream_t stream);
int acattn_launch_adam_step(const acattn_adam_group& g, double lr, double beta1, double beta2, double eps,
                            double weight_decay, int* done, hipStream_t stream);
