// Fused calibrated attention forward, 48 < L <= 64: the hot-path kernel with asynchronous staging.
//
// acattn_fwd_fast.hip stages K, Ka, V and the gate logits through registers and starts computing only when ALL of
// them (29 KB per (sequence, head)) have arrived: the load phase (~10 us at B=512) and the compute phase do not
// overlap, because B*heads = 1024 workgroups are exactly one resident generation on 256 CUs.  Here the staging is
// LDS-DMA (global_load_lds_dwordx4: HBM -> LDS without passing through registers, asynchronous) and is waited
// for in two steps:
//
//     issue:  [q, qa fragments -> registers] [affine weights] [K] [Ka]   |   [V] [G]
//     wait 1: everything left of the bar (s_waitcnt vmcnt(#V + #G instructions): the counter retires in order)
//             -> key halves of the spatial affines, key mask -> pass 1 (scores, spatial calibrator, softmaxes)
//     wait 2: vmcnt(0) + barrier right before pass 2, the first consumer of V and the first global store
//
// so only 13 KB are on the critical path and V / G travel under pass 1.  LDS image and block body are the fast
// kernel's (acattn_fwd_body.inc).
//
// A DMA wave-instruction writes 64 consecutive 16-byte pieces (1 KiB of LDS); piece p of an array lands at byte
// 16*p = row p / PPR, chunk p % PPR (PPR pieces per padded row).  Each lane supplies the GLOBAL address of its
// piece.  Pad pieces and rows past the sequence end are filled from a clamped in-range address: they are never
// read unmasked (their keys carry a -inf mask, their probabilities are exactly 0), any finite data does.
// Every wave issues the SAME number of DMA instructions (surplus slots repeat an earlier chunk: same bytes to the
// same place), so the vmcnt immediates are compile-time constants.
#include <stdlib.h>

#include <type_traits>

#include "acattn_common.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// One wave-wide LDS-DMA: lane l copies the 16 bytes at gsrc (per lane) to LDS byte address lds_dst + 16 * l
// (lds_dst wave-uniform, in M0).  Inline assembly on purpose: the compiler's wait-count insertion treats a
// *known* LDS-DMA as "LDS is being written" and puts s_waitcnt vmcnt(0) in front of the next LDS read of ANY
// address, and it counts a known load into every vmcnt(k) it emits -- both would serialise what this kernel
// overlaps.  The kernel orders the copies itself.
__device__ __forceinline__ void dma16(const float* gsrc, float* lds_dst) {
  const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds_dst;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}

// Workgroup barrier that orders LDS only (__syncthreads() would also drain every outstanding vector-memory op).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Uniform scalars through the scalar cache (lgkmcnt), not the vector-memory pipe: the compiler turns P.scalar[0]
// into a global_load + s_waitcnt vmcnt(0), which would wait for the V / G copies in front of pass 1.
__device__ __forceinline__ float sload_f32(const float* p) {
  float v;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}
__device__ __forceinline__ uint64_t sload_u64(const uint64_t* p) {
  uint64_t v;
  asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
  return v;
}

// Diagnostic builds only (-DACATTN_STAMPS, tools/probe): s_memtime stamps kept in scalar registers and written
// once at the end to the buffer passed in P.noise (unused by this kernel otherwise).  No stamp executes in the
// product build.
#ifdef ACATTN_STAMPS
#define ACATTN_STAMP(k)                                                                             \
  do {                                                                                              \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_[k])::"memory");               \
    __builtin_amdgcn_sched_barrier(0);                                                              \
  } while (0)
#else
#define ACATTN_STAMP(k)
#endif

// wait 2, expanded inside the block body after pass 1
#define ACATTN_BODY_BEFORE_STORES                    \
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
  lds_barrier();

#ifndef ACATTN_DMA_WAVES
#define ACATTN_DMA_WAVES 4  // waves per SIMD the register allocation must leave room for
#endif
template <int DH, bool ADV>
__global__ void __launch_bounds__(256, ACATTN_DMA_WAVES) acattn_fwd_dma_kernel(const acattn_problem P, const acattn_fwd_out O) {
  constexpr int KS = DH / 4;
  constexpr int DT = DH / 16;
  constexpr int VS = DH + 4;
  constexpr int PPR = VS / 4;          // 16-byte pieces per padded K/Ka/V row (the last one is padding)
  constexpr int KV_J = (PPR + 3) / 4;  // DMA instructions per wave and array: 64 rows = PPR chunks over 4 waves
  constexpr int G_J = 4;               // L * GPR <= 64 * 16 pieces = 16 chunks over 4 waves
  constexpr int NT = 4;

#ifdef ACATTN_STAMPS
  unsigned long long stamp_[12] = {};
#endif
  ACATTN_STAMP(0);
  const int L = P.L, H = P.H, nh = P.n_heads;
  constexpr int nT = 4, LP = 64;  // launcher: 48 < L <= 64, 4 waves
  const int GS = (L + 3) & ~3;
  const int GPR = GS / 4;  // pieces per gate row
  int b, h;
  decode_block(blockIdx.x, P.B, nh, b, h);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * L;
  const int hoff = h * DH;
  const size_t bh = (size_t)b * nh + h;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int tile_f = LP * VS;
  float* Ks = smem;
  float* Kas = Ks + tile_f;
  float* Vs = Kas + (ADV ? tile_f : 0);
  float* Gs = Vs + tile_f;                 // [L][GS]  (ADV only)
  float* s_co = Gs + (ADV ? L * GS : 0);   // key half of the order affine
  float* s_cd = s_co + LP;                 // key half of the distance affine
  float* s_lt = s_cd + LP;                 // log(|d| + 1), d = -63 .. 63 (two-sided: no |.| per element)
  const float* s_ltc = s_lt + 63;          // centre of the table
  float* s_w = s_lt + 2 * LP;              // w_order [2*DH], w_dist [2*DH]

  const int qb = wave, i0 = qb * 16, i = i0 + c;
  const bool row_ok = i < L;

  // ---- issue, part 1: register prefetches (inline assembly, pinned after wait 1) -------------------------------
  f4 q4[KS / 4], qa4[KS / 4];
  uint32_t valid;
  {
    const float* qp = P.q + (rowbase + (row_ok ? i : 0)) * H + hoff + KS * g;
    const float* qap = ADV ? P.qa + (rowbase + (row_ok ? i : 0)) * H + hoff + KS * g : qp;
    const uint8_t* vp = P.key_valid + rowbase + min(lane, L - 1);  // every wave reads the 64 key flags itself
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q4[s4]) : "v"(qp + 4 * s4) : "memory");
      if (ADV) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qa4[s4]) : "v"(qap + 4 * s4) : "memory");
    }
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(valid) : "v"(vp) : "memory");
  }
  // ---- issue, part 2: DMA of what pass 1 needs (weights, K, Ka) ---------------------------------------------------
  // piece p = 64 * chunk + lane of a K/Ka/V array sits at (row p / PPR, 16-byte column p % PPR); chunk q + 4 is
  // 256 pieces further on.  One multiply-shift division for the wave's first chunk, increments after that
  // (24-bit multiplies: the 32-bit ones run at quarter rate and this code sits in front of every load).
  {
    int off_kv[KV_J], q_kv[KV_J];
    // affine weights: 2*DH floats each = DH/2 pieces each, one (partial) chunk, issued by every wave
    const int wp = min(lane, DH - 1);
    const float* wsrc = wp < DH / 2 ? P.w_order + 4 * wp : P.w_dist + 4 * (wp - DH / 2);
    if (lane < DH) dma16(wsrc, s_w);
    const float* kb = P.k + rowbase * H + hoff;
    const float* kab = ADV ? P.ka + rowbase * H + hoff : kb;
    constexpr uint32_t kMagic = ((1u << 20) + PPR - 1) / PPR;  // exact for p < 1024
    const int p0 = wave * 64 + lane;
    int row = (int)(__umul24((uint32_t)p0, kMagic) >> 20), ch = p0 - row * PPR;
#pragma unroll
    for (int j = 0; j < KV_J; ++j) {
      const bool surplus = wave + 4 * j >= PPR;  // wave-uniform: repeat this wave's previous chunk
      if (j > 0 && surplus) {
        off_kv[j] = off_kv[j - 1];
        q_kv[j] = q_kv[j - 1];
      } else {
        off_kv[j] = (int)__umul24((uint32_t)min(row, L - 1), (uint32_t)H) + 4 * min(ch, DH / 4 - 1);
        q_kv[j] = wave + 4 * j;
      }
      row += 256 / PPR;
      ch += 256 % PPR;
      if (ch >= PPR) {
        ch -= PPR;
        row += 1;
      }
    }
#pragma unroll
    for (int j = 0; j < KV_J; ++j) dma16(kb + off_kv[j], Ks + q_kv[j] * 256);
    if (ADV) {
#pragma unroll
      for (int j = 0; j < KV_J; ++j) dma16(kab + off_kv[j], Kas + q_kv[j] * 256);
    }
  }
  if (threadIdx.x < 2 * LP - 1) s_lt[threadIdx.x] = fast_log((float)(abs((int)threadIdx.x - 63) + 1));

  ACATTN_STAMP(1);
  // ---- wait 1: q, qa, weights, K, Ka (everything issued so far) ------------------------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float qf[KS], qaf[KS];
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    asm volatile("" : "+v"(q4[s4]));
    if (ADV) asm volatile("" : "+v"(qa4[s4]));
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qf[4 * s4 + e] = row_ok ? q4[s4][e] : 0.f;
      qaf[4 * s4 + e] = (ADV && row_ok) ? qa4[s4][e] : 0.f;
    }
  }
  asm volatile("" : "+v"(valid));
  lds_barrier();  // every wave's K / Ka / weight pieces are in LDS
  ACATTN_STAMP(2);
  // ---- key halves of the two spatial affines, key mask; query halves (rank-1 form of layers.py:705-708,718,726) ----
  {
    const float* s_wo = s_w;
    const float* s_wd = s_w + 2 * DH;
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;  // 4 adjacent lanes per key row, 256 threads = 64 rows
    float co = 0.f, cd = 0.f;
#pragma unroll
    for (int d4 = 0; d4 < DH / 16; ++d4) {
      const int col = part * (DH / 4) + 4 * d4;
      const f4 kv = *(const f4*)(Ks + row * VS + col);
      const f4 wo = *(const f4*)(s_wo + DH + col), wd = *(const f4*)(s_wd + DH + col);
      co += kv.x * wo.x + kv.y * wo.y + kv.z * wo.z + kv.w * wo.w;
      cd += kv.x * wd.x + kv.y * wd.y + kv.z * wd.z + kv.w * wd.w;
    }
    co += __shfl_xor(co, 1);
    co += __shfl_xor(co, 2);
    cd += __shfl_xor(cd, 1);
    cd += __shfl_xor(cd, 2);
    if (part == 0) {
      s_co[row] = -kLog2e * co;  // pre-scaled: sigmoid(o) = 1 / (1 + exp2(ao2 + co2))
      s_cd[row] = cd;
    }
    if (ADV && (L & 3) && b == P.B - 1 && threadIdx.x < 4) {  // the gate piece the DMA left out
      const int col = (L & ~3) + threadIdx.x;
      Gs[(L - 1) * GS + col] = col < L ? P.gate_logits[(rowbase + (L - 1)) * L + col] : 0.f;
    }
  }
  float ao = 0.f, ad = 0.f;
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    const f4 a = *(const f4*)(s_w + KS * g + 4 * s4), d = *(const f4*)(s_w + 2 * DH + KS * g + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ao += qf[4 * s4 + e] * a[e];
      ad += qf[4 * s4 + e] * d[e];
    }
  }
  ao = quad_sum(ao) + sload_f32(P.b_order);
  ad = quad_sum(ad) + sload_f32(P.b_dist);
  const float sc = sload_f32(P.scalar);
  lds_barrier();
  ACATTN_STAMP(3);
  // ---- issue, part 3: V and the gate logits, by wave 0 alone ------------------------------------------------------------
  // They are needed by passes 2 and 3 only, so they go out AFTER the data pass 1 waits for has landed: chip-wide the
  // first 60 % of the input bytes are then exactly the critical ones, and these travel under pass 1.  A wave that
  // issues a DMA while the memory pipe is backed up stalls at the issue (measured: ~3k cycles for these 19
  // instructions), so ONE wave does it: wave 0, whose causal query block has the fewest key tiles.
  if (wave == 0) {
    const float* vb = P.v + rowbase * H + hoff;
    constexpr uint32_t kMagic = ((1u << 20) + PPR - 1) / PPR;
    int row = (int)(__umul24((uint32_t)lane, kMagic) >> 20), ch = lane - row * PPR;
#pragma unroll
    for (int q = 0; q < PPR; ++q) {  // LP * PPR pieces = PPR chunks
      dma16(vb + (int)__umul24((uint32_t)min(row, L - 1), (uint32_t)H) + 4 * min(ch, DH / 4 - 1), Vs + q * 256);
      row += 64 / PPR;
      ch += 64 % PPR;
      if (ch >= PPR) {
        ch -= PPR;
        row += 1;
      }
    }
    if (ADV) {
      const float* gb = P.gate_logits + rowbase * L;
      const int g_pieces = L * GPR;
      const uint32_t g_magic = ((1u << 20) + GPR - 1) / GPR;  // exact for p < 1024, GPR <= 16
      // the row's last piece may run into the next row (finite logits in pad columns are harmless); the one piece
      // that would run past the END of the tensor is left out here and written by hand (below, before pass 1)
      const int tail_p = ((L & 3) && b == P.B - 1) ? g_pieces - 1 : -1;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int p = q * 64 + lane;
        if (q * 64 < g_pieces) {  // wave-uniform
          const int grow = (int)(__umul24((uint32_t)p, g_magic) >> 20), gch = p - (int)__umul24((uint32_t)grow, (uint32_t)GPR);
          if (p < g_pieces && p != tail_p) dma16(gb + (int)__umul24((uint32_t)grow, (uint32_t)L) + 4 * gch, Gs + q * 256);
        }
      }
    }
  }

  const unsigned long long valid_keys = __ballot(lane < L && (valid & 0xFFu) != 0u);
  const int first_valid = valid_keys ? __ffsll((long long)valid_keys) - 1 : L;
  const bool causal = P.causal != 0;
  const int nt_valid = valid_keys ? ((63 - __clzll((long long)valid_keys)) >> 4) + 1 : nT;
  const bool rows_see_a_key = causal ? first_valid <= i0 : valid_keys != 0;
  const int nt = rows_see_a_key ? min(causal ? min(nT, qb + 1) : nT, nt_valid) : nT;

  const float hs2 = 0.5f * (sc * sc);
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const float scale2 = inv_sqrt * kLog2e;
  const bool has_drop = P.p_drop > 0.f;
  const float keep_scale = has_drop ? 1.0f / (1.0f - P.p_drop) : 1.0f;
  const uint32_t prow = ((uint32_t)bh * L + (row_ok ? i : 0)) * (uint32_t)L;
  const uint32_t rng_row = (uint32_t)(bh * L + i);
  const uint64_t seed_eff = P.seed + (P.seed_device ? sload_u64(P.seed_device) : 0ull);
  const RngKey rkey = rng_key(seed_eff);

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

  const float ao2 = -kLog2e * ao;
  const float nc2 = -(hs2 * scale2);

#include "acattn_fwd_body.inc"
  switch (nt) {
    case 1: run_body(std::integral_constant<int, 1>{}); break;
    case 2: run_body(std::integral_constant<int, 2>{}); break;
    case 3: run_body(std::integral_constant<int, 3>{}); break;
    default: run_body(std::integral_constant<int, 4>{}); break;
  }
#ifdef ACATTN_STAMPS
  ACATTN_STAMP(9);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ACATTN_STAMP(10);
  if (lane == 0 && P.noise) {
    unsigned long long* sb = (unsigned long long*)P.noise + ((size_t)blockIdx.x * 4 + wave) * 16;
#pragma unroll
    for (int k = 0; k < 11; ++k) sb[k] = stamp_[k];
  }
#endif
}

template <int DH>
int launch_dma(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  const int LP = 64;
  const int GS = (p.L + 3) & ~3;
  const size_t lds =
      (size_t)((p.adversarial ? 3 : 2) * LP * (DH + 4) + (p.adversarial ? p.L * GS : 0) + 4 * LP + 4 * DH) * sizeof(float);
  const dim3 grid(p.B * p.n_heads), block(256);
  if (p.adversarial) {
    auto k = acattn_fwd_dma_kernel<DH, true>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, stream, p, o);
  } else {
    auto k = acattn_fwd_dma_kernel<DH, false>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, block, lds, stream, p, o);
  }
  return (int)hipGetLastError();
}

}  // namespace

// Returns -100 when the problem is outside this kernel's domain (the caller then tries the register-staged one).
int acattn_launch_fwd_dma(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  static const bool enabled = getenv("ACATTN_DMA") ? atoi(getenv("ACATTN_DMA")) != 0 : true;
  const bool ok = enabled && p.L > 48 && p.L <= 64 && (int64_t)p.B * p.n_heads * p.L * p.L < (1LL << 30) &&
                  (int64_t)p.B * p.L * p.H < (1LL << 30) && p.mask_mode == ACATTN_MASK_STRUCTURED &&
                  p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist &&
                  (!p.adversarial || (p.combine_option == ACATTN_COMBINE_GATE && p.two_level)) && !o.after_spatial &&
                  !o.before_spatial && !o.perturbed_attention && !o.calibrated_attention;
  if (!ok) return -100;
  switch (p.H / p.n_heads) {
    case 16: return launch_dma<16>(p, o, stream);
    case 32: return launch_dma<32>(p, o, stream);
    case 64: return launch_dma<64>(p, o, stream);
  }
  return -100;
}
