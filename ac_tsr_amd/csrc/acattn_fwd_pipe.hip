// Fused calibrated attention forward, software-pipelined: each workgroup walks SEVERAL (sequence, head) items and
// prefetches the next one into a second LDS buffer with LDS-DMA (global_load_lds_dwordx4: no registers, fully
// asynchronous) while the matrix cores and the VALU work on the current one.
//
// Why: with one item per workgroup (acattn_fwd_fast.hip) B*heads = 1024 items are exactly one resident wave of
// workgroups on 256 CUs, so all of them load (HBM busy, ALUs idle), then all compute (ALUs busy, HBM idle), then
// all drain their stores -- the three phases add up instead of overlapping.  Here a
// workgroup's loads for item n+1 are in flight during the whole compute of item n.
//
// LDS image: identical to the fast kernel's (rows padded to DH+4 floats, gate rows to a multiple of 4), so the
// block body (acattn_fwd_body.inc) is shared.  A DMA wave-instruction writes 64 consecutive 16-byte pieces; piece
// p of an array lands at byte 16*p, i.e. row p / PPR, chunk p % PPR (PPR = pieces per padded row).  Each lane
// supplies the GLOBAL address of its piece.
#include <type_traits>

#include "acattn_common.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

// One wave-wide LDS-DMA: lane l copies the 16 bytes at gsrc (per lane) to LDS byte address lds_dst + 16 * l
// (lds_dst wave-uniform, in M0).  Written as inline assembly on purpose: the compiler's wait-count insertion
// treats a *known* LDS-DMA as "LDS is being written" and puts s_waitcnt vmcnt(0) in front of the next LDS read of
// ANY address -- which would serialise the prefetch with the compute it is meant to overlap.  The kernel orders
// the copy itself (s_waitcnt vmcnt(0) + workgroup barrier before the buffer is read).
__device__ __forceinline__ void dma16(const float* gsrc, float* lds_dst) {
  const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)lds_dst;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}

// Workgroup barrier that orders LDS only.  __syncthreads() also drains every outstanding global store of the
// wave (release fence), which here would stall each hand-over on the context stores of the finished item.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Expanded inside the block body after pass 1 (the scores): from here on the wave issues global stores, so this
// is the last point where "all my outstanding vector-memory operations" still means "my prefetches".
#define ACATTN_BODY_BEFORE_STORES asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

template <int DH, bool ADV>
__global__ void __launch_bounds__(256, 2) acattn_fwd_pipe_kernel(const acattn_problem P, const acattn_fwd_out O,
                                                                 const int n_items) {
  constexpr int KS = DH / 4;
  constexpr int DT = DH / 16;
  constexpr int VS = DH + 4;
  constexpr int PPR = VS / 4;  // 16-byte pieces per padded K/Ka/V row (the last one is padding)
  constexpr int NT = 4;

  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;
  const int LP = nT * 16;
  const int GS = (L + 3) & ~3;
  const int GPR = GS / 4;  // pieces per gate row
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n_waves = blockDim.x >> 6;
  const int c = lane & 15, g = lane >> 4;

  // two item buffers + what all items share
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tile_f = LP * VS;
  const int buf_f = (ADV ? 3 : 2) * tile_f + (ADV ? L * GS : 0) + 3 * LP;  // K, [Ka], V, [G], co, cd, km
  float* s_lt = smem + 2 * buf_f;  // log(d + 1): the same for every item

  if (threadIdx.x < LP) s_lt[threadIdx.x] = logf((float)(threadIdx.x + 1));

  const bool causal = P.causal != 0;
  const float sc = P.scalar[0];
  const float hs2 = 0.5f * (sc * sc);
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const float scale2 = inv_sqrt * kLog2e;
  const float nc2 = -(hs2 * scale2);
  const bool has_drop = P.p_drop > 0.f;
  const float keep_scale = has_drop ? 1.0f / (1.0f - P.p_drop) : 1.0f;
  const uint64_t seed_eff = P.seed + (P.seed_device ? *P.seed_device : 0ull);
  const float b_o = P.b_order[0], b_d = P.b_dist[0];
  // both affine weight vectors (query half, key half) live in LDS: registers are the scarce resource here
  float* s_wo = s_lt + LP;      // [2 * DH]
  float* s_wd = s_wo + 2 * DH;  // [2 * DH]
  if (threadIdx.x < 2 * DH) {
    s_wo[threadIdx.x] = P.w_order[threadIdx.x];
    s_wd[threadIdx.x] = P.w_dist[threadIdx.x];
  }

  auto item_bh = [&](int item, int& b, int& h) { decode_block(item, P.B, nh, b, h); };

  // ---- asynchronous staging of one item into buffer `buf` (issue only) --------------------------------------
  // Which 16-byte piece of the source a lane copies does not depend on the item, so the per-lane source offsets
  // are computed once.  Pad pieces and rows past the sequence end are never read unmasked (their keys carry a
  // -inf mask, their probabilities are exactly 0), so they are filled from a clamped in-range address: any
  // finite data does.  Chunk q (64 pieces = 1 KiB of LDS) of an array is issued by wave q % 4.
  constexpr int KV_J = (PPR + 3) / 4;  // chunks of one K/Ka/V array per wave: LP * PPR / 64 = PPR chunks, 4 waves
  constexpr int G_J = 4;               // L * GPR <= 64 * 16 pieces = 16 chunks
  int off_kv[KV_J], off_g[G_J];
#pragma unroll
  for (int j = 0; j < KV_J; ++j) {
    const int p = (wave + 4 * j) * 64 + lane;
    const int row = min(p / PPR, L - 1), ch = min(p % PPR, DH / 4 - 1);
    off_kv[j] = row * H + 4 * ch;
  }
  const int g_pieces = ADV ? L * GPR : 0;
  const int g_tail_off = (L - 1) * L + 4 * (GPR - 1);  // the one piece that can run past the end of the tensor
#pragma unroll
  for (int j = 0; j < G_J; ++j) {
    const int p = (wave + 4 * j) * 64 + lane;
    const int row = p / GPR, ch = p - row * GPR;
    off_g[j] = p < g_pieces ? row * L + 4 * ch : -1;  // -1: lane idle in this chunk
  }
  auto stage = [&](int item, float* buf) {
    int b, h;
    item_bh(item, b, h);
    const size_t base = (size_t)b * L * H + h * DH;
    const float* kb = P.k + base;
    const float* vb = P.v + base;
    const float* kab = ADV ? P.ka + base : nullptr;
#pragma unroll
    for (int j = 0; j < KV_J; ++j) {
      const int q = wave + 4 * j;
      if (q < PPR) {
        dma16(kb + off_kv[j], buf + q * 256);
        if (ADV) dma16(kab + off_kv[j], buf + tile_f + q * 256);
        dma16(vb + off_kv[j], buf + (ADV ? 2 : 1) * tile_f + q * 256);
      }
    }
    if (ADV) {
      float* Gs = buf + 3 * tile_f;
      const float* gb = P.gate_logits + (size_t)b * L * L;
      const int back = (b == P.B - 1 && (L & 3)) ? 4 - (L & 3) : 0;  // keep the tensor's last piece in bounds
#pragma unroll
      for (int j = 0; j < G_J; ++j) {
        const int q = wave + 4 * j;
        if (q * 64 < g_pieces) {
          const int o = off_g[j] == g_tail_off ? off_g[j] - back : off_g[j];  // (patched in finalize())
          if (off_g[j] >= 0) dma16(gb + o, Gs + q * 256);  // idle lanes of the last chunk write nothing
        }
      }
    }
  };

  // ---- after the DMA has landed: key-side calibrator terms and the key mask of the item ------------------------
  const int part = threadIdx.x & 3;  // finalize(): 4 adjacent lanes per key row, DH/4 columns each
  auto finalize = [&](int item, float* buf, uint8_t valid_byte) {
    float* Ks = buf;
    float* s_co = buf + (ADV ? 3 : 2) * tile_f + (ADV ? L * GS : 0);
    float* s_cd = s_co + LP;
    float* s_km = s_cd + LP;
    for (int idx = threadIdx.x; idx < LP * 4; idx += blockDim.x) {
      const int row = idx >> 2;  // blockDim is a multiple of 4: idx & 3 == part
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
        s_co[row] = -kLog2e * co;
        s_cd[row] = cd;
      }
    }
    if (threadIdx.x < LP) {
      float km = ACATTN_NEG_INF;
      if (threadIdx.x < L) km = valid_byte ? 0.f : ACATTN_MASK_FILL * kLog2e;
      s_km[threadIdx.x] = km;
    }
    if (ADV && (L & 3)) {  // the gate piece stage() could not read: last row of the last sequence
      int b, h;
      item_bh(item, b, h);
      if (b == P.B - 1 && threadIdx.x < (L & 3)) {
        const int col = (L & ~3) + threadIdx.x;
        float* Gs = buf + 3 * tile_f;
        Gs[(L - 1) * GS + col] = P.gate_logits[((size_t)b * L + (L - 1)) * L + col];
      }
    }
  };

  // registers prefetched for an item: the wave's query fragments and the thread's key-validity byte
  const int qb = wave, i0 = qb * 16, i = i0 + c;
  const bool row_ok = i < L;
  // Prefetches into registers are inline assembly for the same reason as dma16(): a load the compiler knows about
  // gets a compiler-placed s_waitcnt vmcnt(k) with k counted over the loads IT knows, and since the counter
  // retires in order such a wait also blocks on the (uncounted, later) LDS-DMAs.  The values are "pinned"
  // (pin_regs) after the kernel's own s_waitcnt vmcnt(0) before anything may read them.
  auto fetch_regs = [&](int item, f4 (&q4)[KS / 4], f4 (&qa4)[KS / 4], uint32_t& valid) {
    int b, h;
    item_bh(item, b, h);
    const float* qp = P.q + ((size_t)b * L + (row_ok ? i : 0)) * H + h * DH + KS * g;
    const float* qap = ADV ? P.qa + ((size_t)b * L + (row_ok ? i : 0)) * H + h * DH + KS * g : qp;
    const uint8_t* vp = P.key_valid + (size_t)b * L + min((int)threadIdx.x, L - 1);
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q4[s4]) : "v"(qp + 4 * s4) : "memory");
      if (ADV) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qa4[s4]) : "v"(qap + 4 * s4) : "memory");
    }
    asm volatile("global_load_ubyte %0, %1, off" : "=v"(valid) : "v"(vp) : "memory");
  };
  // after s_waitcnt vmcnt(0): make the prefetched values opaque, then unpack them into fragment order
  auto pin_regs = [&](f4 (&q4)[KS / 4], f4 (&qa4)[KS / 4], uint32_t& valid, float (&qf_)[KS], float (&qaf_)[KS]) {
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      asm volatile("" : "+v"(q4[s4]));
      if (ADV) asm volatile("" : "+v"(qa4[s4]));
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf_[4 * s4 + e] = row_ok ? q4[s4][e] : 0.f;
        qaf_[4 * s4 + e] = (ADV && row_ok) ? qa4[s4][e] : 0.f;
      }
    }
    asm volatile("" : "+v"(valid));
    valid = (threadIdx.x < L) ? (valid & 0xFFu) : 0u;
  };

  int item = blockIdx.x;
  if (item >= n_items) return;
  float qf[KS], qaf[KS];
  f4 q4_n[KS / 4], qa4_n[KS / 4];
  uint32_t valid_n = 0;
  fetch_regs(item, q4_n, qa4_n, valid_n);
  stage(item, smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  pin_regs(q4_n, qa4_n, valid_n, qf, qaf);
  __syncthreads();  // (also publishes s_lt, s_wo, s_wd; drains every compiler-visible load before the loop)
  finalize(item, smem, (uint8_t)valid_n);
  lds_barrier();

  for (int n = 0; item < n_items; ++n) {
    float* buf = smem + (n & 1) * buf_f;
    float* nbuf = smem + ((n + 1) & 1) * buf_f;
    const int next = item + gridDim.x;
    const bool has_next = next < n_items;
    if (has_next) {
      fetch_regs(next, q4_n, qa4_n, valid_n);
      stage(next, nbuf);  // in flight during pass 1 of the compute below
    }

    // ---- compute item `item` from `buf` (same code as the one-item kernel) -------------------------------------
    {
      int b, h;
      item_bh(item, b, h);
      const size_t rowbase = (size_t)b * L;
      const int hoff = h * DH;
      const size_t bh = (size_t)b * nh + h;
      float* Ks = buf;
      float* Kas = Ks + tile_f;
      float* Vs = Kas + (ADV ? tile_f : 0);
      float* Gs = Vs + tile_f;
      float* s_co = buf + (ADV ? 3 : 2) * tile_f + (ADV ? L * GS : 0);
      float* s_cd = s_co + LP;
      float* s_km = s_cd + LP;

      float ao = 0.f, ad = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < KS / 4; ++s4) {
        const f4 a = *(const f4*)(s_wo + KS * g + 4 * s4), d = *(const f4*)(s_wd + KS * g + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          ao += qf[4 * s4 + e] * a[e];
          ad += qf[4 * s4 + e] * d[e];
        }
      }
      ao = quad_sum(ao) + b_o;
      ad = quad_sum(ad) + b_d;
      const float ao2 = -kLog2e * ao;

      const unsigned long long valid_keys = __ballot(lane < L && s_km[lane] == 0.f);
      const int first_valid = valid_keys ? __ffsll((long long)valid_keys) - 1 : L;
      const int nt_valid = valid_keys ? ((63 - __clzll((long long)valid_keys)) >> 4) + 1 : nT;
      const bool rows_see_a_key = causal ? first_valid <= i0 : valid_keys != 0;
      const int nt = rows_see_a_key ? min(causal ? min(nT, qb + 1) : nT, nt_valid) : nT;
      const uint32_t prow = ((uint32_t)bh * L + (row_ok ? i : 0)) * (uint32_t)L;
      const uint32_t rng_row = (uint32_t)(bh * L + i);

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
      auto mask4 = [&](int t) -> f4 {
        const f4 km4 = *(const f4*)(s_km + 16 * t + 4 * g);
        if (causal && (16 * t + 15 > i0)) {
          f4 m;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            m[r] = (16 * t + 4 * g + r > i) ? fminf(km4[r], ACATTN_MASK_FILL * kLog2e) : km4[r];
          return m;
        }
        return km4;
      };
#include "acattn_fwd_body.inc"
      switch (nt) {
        case 1: body(std::integral_constant<int, 1>{}); break;
        case 2: body(std::integral_constant<int, 2>{}); break;
        case 3: body(std::integral_constant<int, 3>{}); break;
        default: body(std::integral_constant<int, 4>{}); break;
      }
    }

    // ---- hand over to the next item --------------------------------------------------------------------------------
    if (has_next) {
      // every wave waited for its own prefetches inside the body (ACATTN_BODY_BEFORE_STORES)
      pin_regs(q4_n, qa4_n, valid_n, qf, qaf);
      lds_barrier();  // all waves: prefetch landed, and nobody reads `buf` any more
      finalize(next, nbuf, (uint8_t)valid_n);
      lds_barrier();
    }
    item = next;
  }
}

template <int DH>
int launch_pipe(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  const int nT = (p.L + 15) / 16, LP = nT * 16;
  const int GS = (p.L + 3) & ~3;
  const int n_items = p.B * p.n_heads;
  const int tile_f = LP * (DH + 4);
  const int buf_f = (p.adversarial ? 3 : 2) * tile_f + (p.adversarial ? p.L * GS : 0) + 3 * LP;
  const size_t lds = (size_t)(2 * buf_f + LP + 4 * DH) * sizeof(float);
  static const int items_per_wg = getenv("ACATTN_PIPE_ITEMS") ? atoi(getenv("ACATTN_PIPE_ITEMS")) : 2;
  const int grid = (n_items + items_per_wg - 1) / items_per_wg;
  if (p.adversarial) {
    auto k = acattn_fwd_pipe_kernel<DH, true>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * nT), lds, stream, p, o, n_items);
  } else {
    auto k = acattn_fwd_pipe_kernel<DH, false>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * nT), lds, stream, p, o, n_items);
  }
  return (int)hipGetLastError();
}

}  // namespace

// Returns -100 when the problem is outside the pipelined path's domain.
int acattn_launch_fwd_pipe(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  static const bool enabled = getenv("ACATTN_PIPE") ? atoi(getenv("ACATTN_PIPE")) != 0 : false;
  const int nT = (p.L + 15) / 16;
  const int dh = p.H / p.n_heads;
  // 4 waves (48 < L <= 64): an array of 64 padded rows is a whole number of 64-piece DMA chunks
  const bool ok = enabled && nT == 4 && (int64_t)p.B * p.n_heads * p.L * p.L < (1LL << 30) &&
                  (int64_t)p.B * p.L * p.H < (1LL << 30) && p.mask_mode == ACATTN_MASK_STRUCTURED &&
                  p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist &&
                  (!p.adversarial || (p.combine_option == ACATTN_COMBINE_GATE && p.two_level)) && !o.after_spatial &&
                  !o.before_spatial && !o.perturbed_attention && !o.calibrated_attention;
  if (!ok) return -100;
  switch (dh) {
    case 16: return launch_pipe<16>(p, o, stream);
    case 32: return launch_pipe<32>(p, o, stream);
    case 64: return launch_pipe<64>(p, o, stream);
  }
  return -100;
}
