// Fused calibrated attention backward -- the training hot path (L <= 64), tuned for gfx950.
//
// Same mathematics as acattn_bwd.hip (the general kernel, which remains the fallback) with the shipped training
// configuration fixed at compile time (structured mask, counter RNG, `gate` combine, both spatial terms,
// two_level) and the work ordered so that only a few row-sized register arrays are alive at any time: the
// general kernel holds ~380 registers per lane (one wave per SIMD); this one stays under 256 (two), with K, Ka
// and V staged once per workgroup in LDS and straight-line code per number of key tiles.
//
// Phases of one 16-row query block (one wave), Pt/Mt = first-level softmaxes before dropout:
//   0  scores on the MFMA, spatial calibrator, Pt and Mt from the saved log-normalisers;
//   1  perturbed branch: dA_p = V.dctx_a, A_p, its softmax backward -> first parts of dP and dM; dv += A_p^T.dctx_a;
//   2  calibrated branch: dA_w = V.dctx_c, A_c, gate, A_w; softmax backward of A_w, gate gradient, softmax
//      backward of A_c -> dP, dM complete; dv += A_w^T.dctx_c;
//   3  external dM, dropout, softmax backward of Mt and Pt -> dSa, dS; spatial-calibrator gradients;
//   4  dq = dS.K (+ rank-1 terms), dqa = dSa.Ka;  dk += dS^T.q, dka += dSa^T.qa.
#include <type_traits>

#include "acattn_common.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, false));
  return v;
}

__device__ __forceinline__ float hsum(const f4 v) { return (v[0] + v[1]) + (v[2] + v[3]); }

// Diagnostic builds only (-DACATTN_BWD_STAMPS, tools/gpu_bwd_stamps.sh): s_memtime per phase and wave, kept in scalar
// registers and written at the end to g_bwd_stamps[wave][16].  No stamp executes in the product build.
#ifdef ACATTN_BWD_STAMPS
__device__ unsigned long long g_bwd_stamps[8192 * 16];
#define BWD_STAMP(k)                                                                  \
  do {                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_[k])::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                                \
  } while (0)
#else
#define BWD_STAMP(k)
#endif

template <int DH>
__global__ void __launch_bounds__(256, 2) acattn_bwd_fast_kernel(const acattn_problem P, const acattn_bwd_io IO) {
  constexpr int KS = DH / 4;
  constexpr int DT = DH / 16;
  constexpr int VS = DH + 4;

  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;
  const int LP = nT * 16;
  const int SS = 16 * (nT | 1);  // scratch row stride: 16 * odd
  int b, h;
  decode_block(blockIdx.x, P.B, nh, b, h);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const bool full = IO.attack_only == 0;  // attack_only: only dqa and dka will be read (acattn.h)
  const size_t rowbase = (size_t)b * L;
  const int hoff = h * DH;
  const size_t bh = (size_t)b * nh + h;
#ifdef ACATTN_BWD_STAMPS
  unsigned long long stamp_[16] = {};
#endif
  BWD_STAMP(0);

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;                // [LP][VS]
  float* Kas = Ks + LP * VS;
  float* Vs = Kas + LP * VS;
  float* aK = Vs + LP * VS;        // key-side accumulators
  float* aKa = aK + LP * VS;
  float* aV = aKa + LP * VS;
  float* s_co = aV + LP * VS;      // -log2e * (k_j . w_order[dh:])
  float* s_cd = s_co + LP;
  float* s_km = s_cd + LP;         // exp2-domain key mask
  float* s_lt = s_km + LP;
  float* s_dco = s_lt + LP;        // column sums of d(order affine) / d(distance affine)
  float* s_dcd = s_dco + LP;
  float* s_dwq = s_dcd + LP;       // [2*DH] query halves of dw_order, dw_dist
  float* s_small = s_dwq + 2 * DH; // [8] db_order, db_dist, dscalar
  int* s_cnt = (int*)(s_small + 8);  // [4 exchanges][4 query blocks]: key tiles the block publishes (0: nothing)
  float* scratch_all = s_small + 8 + 16;
  float* scratch = scratch_all + wave * 16 * SS;  // this wave's published tile set: [16 query rows][SS]

  // ---- query-block fragments (straight from HBM) -------------------------------------------------------
  const int qb = wave, i0 = qb * 16, i = i0 + c;
  const bool row_ok = i < L;
  float qf[KS], qaf[KS];
  // row fragment of a [B,L,H] tensor for this lane (zeros for padding rows / a NULL tensor); the cotangent fragments
  // are fetched where they are used (L2 hits) rather than held in 2 x KS registers through the whole kernel
  auto row_frag = [&](const float* base, float (&dst)[KS]) {
    const size_t off = (rowbase + (row_ok ? i : 0)) * H + hoff + KS * g;
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      f4 t = {0.f, 0.f, 0.f, 0.f};
      if (row_ok && base) t = *(const f4*)(base + off + 4 * s4);
#pragma unroll
      for (int e = 0; e < 4; ++e) dst[4 * s4 + e] = t[e];
    }
  };
  {
    const size_t off = (rowbase + (row_ok ? i : 0)) * H + hoff + KS * g;
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      f4 t = {0.f, 0.f, 0.f, 0.f}, ta = t;
      if (row_ok) {
        t = *(const f4*)(P.q + off + 4 * s4);
        ta = *(const f4*)(P.qa + off + 4 * s4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf[4 * s4 + e] = t[e];
        qaf[4 * s4 + e] = ta[e];
      }
    }
  }
  f4 st0 = {0.f, 0.f, 0.f, 0.f};
  float lse_w2 = 0.f;
  if (row_ok) {
    const float* sp = IO.row_stats + (bh * L + i) * ACATTN_NSTAT;
    st0 = *(const f4*)sp;
    lse_w2 = sp[4] * kLog2e;
  }
  const float lse_x2 = st0[0] * kLog2e, lse_y2 = st0[1] * kLog2e, lse_u2 = st0[2] * kLog2e, lse_v2 = st0[3] * kLog2e;

  // ---- stage K, Ka, V; zero accumulators; key-side calibrator terms ------------------------------------------
  constexpr int KV_IT = DH / 16;
  f4 r_k[KV_IT], r_ka[KV_IT], r_v[KV_IT];
#pragma unroll
  for (int it = 0; it < KV_IT; ++it) {
    const int idx = threadIdx.x + it * blockDim.x;
    const int row = idx / (DH / 4), c4 = idx - row * (DH / 4);
    r_k[it] = f4{0.f, 0.f, 0.f, 0.f};
    r_ka[it] = r_k[it];
    r_v[it] = r_k[it];
    if (row < L) {
      const size_t o = (rowbase + row) * H + hoff + 4 * c4;
      r_k[it] = *(const f4*)(P.k + o);
      r_ka[it] = *(const f4*)(P.ka + o);
      r_v[it] = *(const f4*)(P.v + o);
    }
  }
  const f4 w_ko = *(const f4*)(P.w_order + DH + 4 * (threadIdx.x % (DH / 4)));
  const f4 w_kd = *(const f4*)(P.w_dist + DH + 4 * (threadIdx.x % (DH / 4)));
  const uint8_t r_valid = (threadIdx.x < L) ? P.key_valid[rowbase + threadIdx.x] : (uint8_t)0;
  BWD_STAMP(1);
#pragma unroll
  for (int it = 0; it < KV_IT; ++it) {
    const int idx = threadIdx.x + it * blockDim.x;
    const int row = idx / (DH / 4), c4 = idx - row * (DH / 4);
    const f4 kv = r_k[it];
    *(f4*)(Ks + row * VS + 4 * c4) = kv;
    *(f4*)(Kas + row * VS + 4 * c4) = r_ka[it];
    *(f4*)(Vs + row * VS + 4 * c4) = r_v[it];
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    *(f4*)(aK + row * VS + 4 * c4) = z;
    *(f4*)(aKa + row * VS + 4 * c4) = z;
    *(f4*)(aV + row * VS + 4 * c4) = z;
    float co = kv.x * w_ko.x + kv.y * w_ko.y + kv.z * w_ko.z + kv.w * w_ko.w;
    float cd = kv.x * w_kd.x + kv.y * w_kd.y + kv.z * w_kd.z + kv.w * w_kd.w;
#pragma unroll
    for (int off = 1; off < DH / 4; off <<= 1) {
      co += __shfl_xor(co, off);
      cd += __shfl_xor(cd, off);
    }
    if (c4 == 0) {
      s_co[row] = -kLog2e * co;
      s_cd[row] = cd;
    }
  }
  if (threadIdx.x < LP) {
    const int j = threadIdx.x;
    float km = ACATTN_NEG_INF;
    if (j < L) km = r_valid ? 0.f : ACATTN_MASK_FILL * kLog2e;
    s_km[j] = km;
    s_lt[j] = logf((float)(j + 1));
    s_dco[j] = 0.f;
    s_dcd[j] = 0.f;
  }
  for (int j = threadIdx.x; j < 2 * DH + 8; j += blockDim.x) s_dwq[j] = 0.f;  // s_dwq and s_small are contiguous

  float ao = 0.f, ad = 0.f;
#pragma unroll
  for (int s4 = 0; s4 < KS / 4; ++s4) {
    const f4 a = *(const f4*)(P.w_order + KS * g + 4 * s4), d = *(const f4*)(P.w_dist + KS * g + 4 * s4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ao += qf[4 * s4 + e] * a[e];
      ad += qf[4 * s4 + e] * d[e];
    }
  }
  ao = quad_sum(ao) + P.b_order[0];
  ad = quad_sum(ad) + P.b_dist[0];
  const float sc = P.scalar[0];
  const uint64_t seed_eff = P.seed + (P.seed_device ? *P.seed_device : 0ull);
  const RngKey rkey = rng_key(seed_eff);
  __syncthreads();
  BWD_STAMP(2);

  const unsigned long long valid_keys = __ballot(lane < L && s_km[lane] == 0.f);
  const int first_valid = valid_keys ? __ffsll((long long)valid_keys) - 1 : L;
  const bool causal = P.causal != 0;
  const int nt_valid = valid_keys ? ((63 - __clzll((long long)valid_keys)) >> 4) + 1 : nT;
  const bool rows_see_a_key = causal ? first_valid <= i0 : valid_keys != 0;
  const int nt = rows_see_a_key ? min(causal ? min(nT, qb + 1) : nT, nt_valid) : nT;

  const float s2 = sc * sc;
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const float scale2 = inv_sqrt * kLog2e;
  const float ao2 = -kLog2e * ao;
  const float nc2 = -(0.5f * s2 * scale2);
  const bool has_drop = P.p_drop > 0.f;
  const float keep_scale = has_drop ? 1.0f / (1.0f - P.p_drop) : 1.0f;
  const uint32_t prow = ((uint32_t)bh * L + (row_ok ? i : 0)) * (uint32_t)L;
  const float dpen2 = IO.d_penalty_part ? 2.0f * IO.d_penalty_part[(size_t)bh * nT + qb] : 0.f;  // (uniform per wave)
  const uint32_t rng_row = (uint32_t)(bh * L + i);
  const float okf = row_ok ? 1.0f : 0.0f;

  auto mask4 = [&](int t) -> f4 {
    const f4 km4 = *(const f4*)(s_km + 16 * t + 4 * g);
    if (causal && (16 * t + 15 > i0)) {
      const int dq = (c - 4 * g) + 16 * (qb - t);  // key after query  <=>  dq < r  (see acattn_fwd_dma.hip)
      f4 m;
#pragma unroll
      for (int r = 0; r < 4; ++r) m[r] = (dq < r) ? fminf(km4[r], ACATTN_MASK_FILL * kLog2e) : km4[r];
      return m;
    }
    return km4;
  };
  // sigmoid(order affine), its log-argument and the distance residual of tile t (recomputed where needed)
  auto spatial = [&](int t, f4& pr, f4& val, f4& df) {
    const f4 ea = *(const f4*)(s_co + 16 * t + 4 * g) + ao2;
    const f4 cd4 = *(const f4*)(s_cd + 16 * t + 4 * g);
    const int d0 = i - (16 * t + 4 * g);
    f4 lt4;
    if (16 * t + 15 > i0) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = fast_rcp(1.0f + ex2(ea[r]));
        val[r] = (d0 - r < 0) ? pr[r] : 1.0f - pr[r];
        lt4[r] = s_lt[d0 - r < 0 ? r - d0 : d0 - r];
      }
    } else {
      const float* lp = s_lt + d0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        pr[r] = fast_rcp(1.0f + ex2(ea[r]));
        val[r] = 1.0f - pr[r];
        lt4[r] = lp[-r];
      }
    }
    df = lt4 - (cd4 + ad);
  };
  auto load_seg = [&](const float* base, int t) -> f4 {  // 4 keys of this row from a [B,nh,L,L] / [B,L,L] row
    const int j0 = 16 * t + 4 * g;
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (row_ok && j0 < L) {
      const float* p = base + j0;
      if (j0 + 3 < L) {
        v = *(const f4u*)p;
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (j0 + r < L) v[r] = p[r];
      }
    }
    return v;
  };
  auto store_seg = [&](float* base, int t, const f4 val) {
    const int j0 = 16 * t + 4 * g;
    if (!row_ok || j0 >= L) return;
    float* p = base + j0;
    if (j0 + 3 < L) {
      *(f4u*)p = val;
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j0 + r < L) p[r] = val[r];
    }
  };

  // ---- key side: dv, dk, dka = sum over query rows of tile^T . rows ---------------------------------------------------
  // The query blocks hold the tiles, the sums run over all of them.  LDS float atomics would do it in one line and
  // are ruinous here: ds_add_f32 retires one LANE per ~3 cycles (193 cycles per wave instruction against 4 for a write,
  // tools/probe/lds_atomic.hip), and the 160 of them per wave were half of this kernel's time.  Instead every wave
  // PUBLISHES its tiles in its scratch area, and after a barrier wave w, OWNER of key tile w, forms that tile's sum over
  // the publishing blocks on the MFMA and adds it to the accumulator rows only it touches.  Under the causal mask block
  // qb publishes qb + 1 tiles while owner w reads nT - w blocks: the two triangles add up to equal work per wave.
  // Every wave of the workgroup walks the same exchanges (launch-uniform flags), publishing nothing where its block
  // has nothing (no cotangent, mask-only, ...): the barriers count arrivals.
  const bool ex_att = IO.d_ctx_attacked && full, ex_cal = IO.d_ctx_calibrated && full, ex_k = full;
  bool tiles_in_use = false;  // an earlier exchange's tiles may still be read by their owners
  // rows of the B operand: requested before the barriers (they depend on nobody), one trip for all blocks.  A block
  // whose rows see a valid key publishes at most qb + 1 tiles under the causal mask, so owner `wave` skips the blocks
  // above it (a block that sees none spreads its soft-max over every key, the later ones included).
  auto request_rows = [&](const float* rows, float (&bv)[4][4][DT]) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      if (qq >= nT || (qq < wave && causal && first_valid <= 16 * qq)) continue;  // (uniform) below: never published
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int qi = min(16 * qq + 4 * s + g, L - 1);  // padding rows: their tile entries are zero
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) bv[qq][s][dt] = rows[(rowbase + qi) * H + hoff + 16 * dt + c];
      }
    }
  };
  auto own_tile = [&](int e, const float (&bv)[4][4][DT], float* acc) {
    const int4 cn = *(const int4*)(s_cnt + 4 * e);
    const int cnt[4] = {__builtin_amdgcn_readfirstlane(cn.x), __builtin_amdgcn_readfirstlane(cn.y),
                        __builtin_amdgcn_readfirstlane(cn.z), __builtin_amdgcn_readfirstlane(cn.w)};
    f4 o[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) o[dt] = f4{0.f, 0.f, 0.f, 0.f};
    bool any = false;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      if (qq >= nT || (qq < wave && causal && first_valid <= 16 * qq) || cnt[qq] <= wave) continue;  // (uniform)
      any = true;
      const float* sc = scratch_all + qq * 16 * SS;
      float a[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) a[s] = sc[(4 * s + g) * SS + 16 * wave + c];
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = mfma16(a[s], bv[qq][s][dt], o[dt]);
    }
    if (any) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[(16 * wave + 4 * g + r) * VS + 16 * dt + c] += o[dt][r];  // this wave's rows only
    }
  };
  // one exchange: `publish` writes the block's tiles into `scratch` and returns how many (0: nothing to write)
  auto exchange = [&](int e, const float* rows, float* acc, auto publish) {
    float bv[4][4][DT];
    request_rows(rows, bv);
    if (tiles_in_use) __syncthreads();
    tiles_in_use = true;
    const int n = publish();
    if (lane == 0) s_cnt[4 * e + qb] = n;
    __syncthreads();
    own_tile(e, bv, acc);
  };
  auto nothing = [] { return 0; };

  // MASK_ONLY: a query block none of whose rows carries a CONTEXT cotangent while the attack mask does carry one (the
  // attacked-loss pass through the last layer: the context is read at one position per sequence, the mask penalty
  // reaches every row).  With dA_p = dA_w = 0 every term of the chain vanishes except the soft-max of the mask scores:
  //     dSa = Mt (keep . dM_out - sum_j Mt keep dM_out) / sqrt(dh)  ->  dqa, dka;   dq = dk = dv = dgate = 0
  // so such a block forms Sa only: no spatial calibrator, no noise, none of the five exponentials of the other branches.
  auto body = [&](auto ntb_c, auto mask_only_c) {
    constexpr int NTB = decltype(ntb_c)::value;
    constexpr bool MASK_ONLY = decltype(mask_only_c)::value;

    auto tiles_of = [&](const f4 (&tile)[NTB]) {
      return [&] {
#pragma unroll
        for (int t = 0; t < NTB; ++t) *(f4*)(scratch + c * SS + 16 * t + 4 * g) = tile[t];
        return NTB;
      };
    };
    // MFMA tile:  out[t] = X_tile(t) . frag^T   with X in {K, Ka, V} staged in LDS (A operand) and a row fragment (B)
    auto score_tiles = [&](const float* Xs, const float (&frag)[KS], f4 (&out)[NTB]) {
#pragma unroll
      for (int t = 0; t < NTB; ++t) out[t] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < KS / 4; ++s4) {
#pragma unroll
        for (int t = 0; t < NTB; ++t) {
          const f4 x4 = *(const f4*)(Xs + (16 * t + c) * VS + KS * g + 4 * s4);
#pragma unroll
          for (int e = 0; e < 4; ++e) out[t] = mfma16(x4[e], frag[4 * s4 + e], out[t]);
        }
      }
    };

    // ---- phase 0: Pt, Mt ------------------------------------------------------------------------------------
    f4 tS[NTB], tM[NTB];
    if constexpr (!MASK_ONLY) score_tiles(Ks, qf, tS);
    score_tiles(Kas, qaf, tM);
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
      const f4 mk4 = mask4(t);
      if constexpr (!MASK_ONLY) {
        f4 pr, val, df;
        spatial(t, pr, val, df);
        f4 lg;
#pragma unroll
        for (int r = 0; r < 4; ++r) lg[r] = __builtin_amdgcn_logf(val[r] + ACATTN_LOG_EPS);
        f4 x = tS[t] * scale2 + mk4;
        x = lg * inv_sqrt + x;
        x = (df * df) * nc2 + x;
#pragma unroll
        for (int r = 0; r < 4; ++r) tS[t][r] = ex2(x[r] - lse_x2) * okf;
      }
      const f4 y = tM[t] * scale2 + mk4;
#pragma unroll
      for (int r = 0; r < 4; ++r) tM[t][r] = ex2(y[r] - lse_y2) * okf;
    }

    BWD_STAMP(3);
    // ---- phase 1: perturbed branch ---------------------------------------------------------------------------
    f4 dPa[NTB], dMa[NTB];
    uint32_t keepA = 0, keepM = 0;  // dropout keep bits, 4 per tile
    if constexpr (MASK_ONLY) {
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        const RngGroup rg = rng_group(rkey, rng_row, (uint32_t)(4 * t + g), P.p_drop);
        keepM |= (has_drop ? rg.keep_mask : 0xFu) << (4 * t);
        dMa[t] = f4{0.f, 0.f, 0.f, 0.f};
        dPa[t] = dMa[t];
      }
      if (full)
        for (int t = 0; t < nT; ++t) store_seg(IO.dgate_logits + prow, t, f4{0.f, 0.f, 0.f, 0.f});
    } else if (!IO.d_ctx_attacked) {
      // [r4] no cotangent of the attacked context (every layer but the last: layers.py:1112): d A_p = 0, so the perturbed
      // attention, its product with V and the Gaussian draws that only enter through it are not needed -- the keep bits are
      // (wave-uniform branch; the streaming pair has the same form as a template flag, acattn_bwd_stream.hip)
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        const RngGroup rg = rng_group(rkey, rng_row, (uint32_t)(4 * t + g), P.p_drop);
        keepA |= (has_drop ? rg.keep_after : 0xFu) << (4 * t);
        keepM |= (has_drop ? rg.keep_mask : 0xFu) << (4 * t);
        dMa[t] = f4{0.f, 0.f, 0.f, 0.f};
        dPa[t] = dMa[t];
      }
    } else {
      f4 dAp[NTB], Ap[NTB], nz[NTB];
      float gaf[KS];
      row_frag(IO.d_ctx_attacked, gaf);
      score_tiles(Vs, gaf, dAp);
      float da = 0.f;
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        const RngGroup rg = rng_group(rkey, rng_row, (uint32_t)(4 * t + g), P.p_drop);
        nz[t] = rg.n;
        keepA |= (has_drop ? rg.keep_after : 0xFu) << (4 * t);
        keepM |= (has_drop ? rg.keep_mask : 0xFu) << (4 * t);
        const f4 sa = has_drop ? keep_scale4(rg.keep_after, keep_scale) : f4{1.f, 1.f, 1.f, 1.f};
        const f4 sm = has_drop ? keep_scale4(rg.keep_mask, keep_scale) : f4{1.f, 1.f, 1.f, 1.f};
        const f4 p = tS[t] * sa, m = tM[t] * sm;
        const f4 u2 = (p * m + nz[t] * (1.0f - m)) * kLog2e + (mask4(t) - lse_u2);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ap[t][r] = ex2(u2[r]) * okf;
        da += hsum(Ap[t] * dAp[t]);
      }
      da = quad_sum(da);
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        f4 sa, sm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sa[r] = ((keepA >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
          sm[r] = ((keepM >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
        }
        const f4 p = tS[t] * sa, m = tM[t] * sm;
        const f4 du = Ap[t] * (dAp[t] - da);
        dPa[t] = du * m;
        dMa[t] = du * (p - nz[t]);
      }
      if (ex_att) exchange(0, IO.d_ctx_attacked, aV, tiles_of(Ap));
    }

    BWD_STAMP(4);
    // ---- phase 2: calibrated branch ----------------------------------------------------------------------------
    if constexpr (!MASK_ONLY) {
      f4 dAw[NTB], Ac[NTB], Aw[NTB];
      float gcf[KS];
      row_frag(IO.d_ctx_calibrated, gcf);
      score_tiles(Vs, gcf, dAw);
      const float* grow = P.gate_logits + (rowbase + (row_ok ? i : 0)) * L;
      float dc = 0.f;
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        f4 sa, sm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sa[r] = ((keepA >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
          sm[r] = ((keepM >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
        }
        const f4 p = tS[t] * sa, m = tM[t] * sm;
        const f4 mk4 = mask4(t);
        const f4 a1 = m * (-kLog2e) + kLog2e;
        f4 ex1;
#pragma unroll
        for (int r = 0; r < 4; ++r) ex1[r] = ex2(a1[r]);
        const f4 v2 = (p * ex1) * kLog2e + (mk4 - lse_v2);
        const f4 gt = gate_value(load_seg(grow, t), P.gate_is_prob);
#pragma unroll
        for (int r = 0; r < 4; ++r) Ac[t][r] = ex2(v2[r]) * okf;
        const f4 w2 = (gt * (p - Ac[t]) + Ac[t]) * kLog2e + (mk4 - lse_w2);
#pragma unroll
        for (int r = 0; r < 4; ++r) Aw[t][r] = ex2(w2[r]) * okf;
        dc += hsum(Aw[t] * dAw[t]);
      }
      dc = quad_sum(dc);
      float r1 = 0.f;
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        f4 sa;
#pragma unroll
        for (int r = 0; r < 4; ++r) sa[r] = ((keepA >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
        const f4 p = tS[t] * sa;
        const f4 gt = gate_value(load_seg(grow, t), P.gate_is_prob);
        const f4 dw = Aw[t] * (dAw[t] - dc);                       // = d A_g
        if (full) store_seg(IO.dgate_logits + prow, t, dw * (p - Ac[t]) * (gt * (1.0f - gt)));
        dPa[t] += gt * dw;
        const f4 dac = (1.0f - gt) * dw;
        dAw[t] = dac;                                              // reuse: d A_c
        r1 += hsum(Ac[t] * dac);
      }
      if (full)
        for (int t = NTB; t < nT; ++t) store_seg(IO.dgate_logits + prow, t, f4{0.f, 0.f, 0.f, 0.f});
      if (ex_cal) exchange(1, IO.d_ctx_calibrated, aV, tiles_of(Aw));
      r1 = quad_sum(r1);
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        f4 sa, sm;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sa[r] = ((keepA >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
          sm[r] = ((keepM >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
        }
        const f4 p = tS[t] * sa, m = tM[t] * sm;
        const f4 a1 = m * (-kLog2e) + kLog2e;
        f4 ex1;
#pragma unroll
        for (int r = 0; r < 4; ++r) ex1[r] = ex2(a1[r]);
        const f4 dv = Ac[t] * (dAw[t] - r1);
        dPa[t] += dv * ex1;
        dMa[t] -= dv * (p * ex1);
      }
    }

    BWD_STAMP(5);
    // ---- phase 3: through the first-level softmaxes; spatial-calibrator gradients ------------------------------
    float r2 = 0.f, r3 = 0.f;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
      if (IO.d_attack_mask) dMa[t] += load_seg(IO.d_attack_mask + prow, t);
      f4 sa, sm;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sa[r] = ((keepA >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
        sm[r] = ((keepM >> (4 * t + r)) & 1u) ? keep_scale : 0.f;
      }
      // the mask penalty's cotangent from the rebuilt tile (acattn_bwd_io.d_penalty_part): d M = 2 d_pen (M - 1), M = Mt . keep
      if (IO.d_penalty_part) dMa[t] += (tM[t] * sm - 1.0f) * dpen2;
      dMa[t] *= sm;
      r2 += hsum(tM[t] * dMa[t]);
      if constexpr (!MASK_ONLY) {
        dPa[t] *= sa;
        r3 += hsum(tS[t] * dPa[t]);
      }
    }
    r2 = quad_sum(r2);
    if constexpr (!MASK_ONLY) r3 = quad_sum(r3);
    float da_o = 0.f, da_d = 0.f, dsc = 0.f;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
      dMa[t] = (tM[t] * (dMa[t] - r2)) * inv_sqrt;  // dSa
      if constexpr (MASK_ONLY) continue;
      dPa[t] = (tS[t] * (dPa[t] - r3)) * inv_sqrt;  // dS (the calibrator terms are additive)
      f4 pr, val, df;
      spatial(t, pr, val, df);
      f4 d_o, d_d;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sgn = (16 * t + 4 * g + r > i) ? 1.0f : -1.0f;
        d_o[r] = dPa[t][r] * (sgn * pr[r] * (1.0f - pr[r])) * fast_rcp(val[r] + ACATTN_LOG_EPS);
      }
      d_d = dPa[t] * (df * s2);
      dsc += hsum(dPa[t] * (df * df)) * (-sc);
      da_o += hsum(d_o);
      da_d += hsum(d_d);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float so = row16_sum(d_o[r]), sd = row16_sum(d_d[r]);
        if (c == 0) {
          atomicAdd(s_dco + 16 * t + 4 * g + r, so);
          atomicAdd(s_dcd + 16 * t + 4 * g + r, sd);
        }
      }
    }
    da_o = quad_sum(da_o);
    da_d = quad_sum(da_d);
    dsc = quad_sum(dsc);
    if constexpr (!MASK_ONLY) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float vo = row16_sum(da_o * qf[s]), vd = row16_sum(da_d * qf[s]);
        if (c == 0) {
          atomicAdd(s_dwq + KS * g + s, vo);
          atomicAdd(s_dwq + DH + KS * g + s, vd);
        }
      }
    }
    if constexpr (!MASK_ONLY) {
      const float so = row16_sum(da_o), sd = row16_sum(da_d), ss = row16_sum(dsc);  // totals over the 16 rows
      if (lane == 0) {
        atomicAdd(s_small + 0, so);
        atomicAdd(s_small + 1, sd);
        atomicAdd(s_small + 2, ss);
      }
    }

    BWD_STAMP(6);
    // ---- phase 4: dq, dqa; dk, dka ------------------------------------------------------------------------------
    {
      f4 oq[DT], oqa[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        oq[dt] = f4{0.f, 0.f, 0.f, 0.f};
        oqa[dt] = oq[dt];
      }
#pragma unroll
      for (int t = 0; t < NTB; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* kp = Ks + (16 * t + 4 * g + r) * VS + c;
          const float* kap = Kas + (16 * t + 4 * g + r) * VS + c;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            if constexpr (!MASK_ONLY)
              if (full) oq[dt] = mfma16(kp[16 * dt], dPa[t][r], oq[dt]);
            oqa[dt] = mfma16(kap[16 * dt], dMa[t][r], oqa[dt]);
          }
        }
      if (row_ok) {
        const uint32_t off = ((uint32_t)rowbase + i) * H + hoff + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          // rank-1 terms of dq: query halves of the affine weights in the lane's output-column order
          const f4 wo_lo = *(const f4*)(P.w_order + 16 * dt + 4 * g), wd_lo = *(const f4*)(P.w_dist + 16 * dt + 4 * g);
          if (full) *(f4*)(IO.dq + off + 16 * dt) = oq[dt] + da_o * wo_lo + da_d * wd_lo;
          *(f4*)(IO.dqa + off + 16 * dt) = oqa[dt];
        }
      }
    }
    BWD_STAMP(7);
    if constexpr (MASK_ONLY) {  // the exchanges this block has nothing for
      if (ex_att) exchange(0, IO.d_ctx_attacked, aV, nothing);
      if (ex_cal) exchange(1, IO.d_ctx_calibrated, aV, nothing);
      if (ex_k) exchange(2, P.q, aK, nothing);
    } else {
      if (ex_k) exchange(2, P.q, aK, tiles_of(dPa));
    }
    exchange(3, P.qa, aKa, tiles_of(dMa));
    BWD_STAMP(8);
  };

  // A query block none of whose rows carries a cotangent contributes nothing anywhere: its wave only writes the zeros
  // the caller expects in dq, dqa and the gate partials (the last layer of the models is read at one position per
  // sequence, so three of its four blocks are such blocks in the calibrated-loss pass).
  const bool block_active = qblock_active(IO, b, qb);
  if (block_active && !qblock_has_ctx(IO, b, qb)) {  // only the mask cotangent reaches this block
    switch (nt) {
      case 1: body(std::integral_constant<int, 1>{}, std::true_type{}); break;
      case 2: body(std::integral_constant<int, 2>{}, std::true_type{}); break;
      case 3: body(std::integral_constant<int, 3>{}, std::true_type{}); break;
      default: body(std::integral_constant<int, 4>{}, std::true_type{}); break;
    }
  } else if (block_active) {
    switch (nt) {
      case 1: body(std::integral_constant<int, 1>{}, std::false_type{}); break;
      case 2: body(std::integral_constant<int, 2>{}, std::false_type{}); break;
      case 3: body(std::integral_constant<int, 3>{}, std::false_type{}); break;
      default: body(std::integral_constant<int, 4>{}, std::false_type{}); break;
    }
  } else {
    if (row_ok) {
      const f4 z = {0.f, 0.f, 0.f, 0.f};
      const uint32_t off = ((uint32_t)rowbase + i) * H + hoff + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        if (full) *(f4*)(IO.dq + off + 16 * dt) = z;
        *(f4*)(IO.dqa + off + 16 * dt) = z;
      }
      if (full)
        for (int t = 0; t < nT; ++t) store_seg(IO.dgate_logits + prow, t, z);
    }
    // no tiles of its own, but the owner of key tile `wave` all the same
    if (ex_att) exchange(0, IO.d_ctx_attacked, aV, nothing);
    if (ex_cal) exchange(1, IO.d_ctx_calibrated, aV, nothing);
    if (ex_k) exchange(2, P.q, aK, nothing);
    exchange(3, P.qa, aKa, nothing);
  }

  // ---- key-side results and parameter partials ---------------------------------------------------------------------
  __syncthreads();
  BWD_STAMP(9);
  for (int idx = threadIdx.x; idx < L * (DH / 4); idx += blockDim.x) {
    const int row = idx / (DH / 4), c4 = idx - row * (DH / 4);
    const f4 r1k = s_dco[row] * *(const f4*)(P.w_order + DH + 4 * c4) + s_dcd[row] * *(const f4*)(P.w_dist + DH + 4 * c4);
    const size_t o = (rowbase + row) * H + hoff + 4 * c4;
    *(f4*)(IO.dka + o) = *(const f4*)(aKa + row * VS + 4 * c4);
    if (full) {
      *(f4*)(IO.dk + o) = *(const f4*)(aK + row * VS + 4 * c4) + r1k;
      *(f4*)(IO.dv + o) = *(const f4*)(aV + row * VS + 4 * c4);
    }
  }
  if (!full) return;  // the parameter partials are not read either
  for (int d = threadIdx.x; d < 2 * DH; d += blockDim.x) {
    float wo = 0.f, wd = 0.f;
    if (d < DH) {
      wo = s_dwq[d];
      wd = s_dwq[DH + d];
    } else {
      for (int j = 0; j < L; ++j) {
        const float kv = Ks[j * VS + (d - DH)];
        wo += s_dco[j] * kv;
        wd += s_dcd[j] * kv;
      }
    }
    IO.dw_order_part[bh * (IO.part_stride ? IO.part_stride : 2 * DH) + d] = wo;
    IO.dw_dist_part[bh * (IO.part_stride ? IO.part_stride : 2 * DH) + d] = wd;
  }
  if (threadIdx.x < 4)
    IO.dsmall_part[bh * (IO.part_stride ? IO.part_stride : 4) + threadIdx.x] = threadIdx.x < 3 ? s_small[threadIdx.x] : 0.f;
#ifdef ACATTN_BWD_STAMPS
  BWD_STAMP(10);
  if (lane == 0 && blockIdx.x * 4 + wave < 8192) {
    stamp_[11] = (unsigned long long)qb;
    for (int k = 0; k < 12; ++k) g_bwd_stamps[(blockIdx.x * 4 + wave) * 16 + k] = stamp_[k];
  }
#endif
}

template <int DH>
int launch_fast(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  const int nT = (p.L + 15) / 16, LP = nT * 16, SS = 16 * (nT | 1);
  const size_t lds = (size_t)(6 * LP * (DH + 4) + 6 * LP + 2 * DH + 8 + 16 + nT * 16 * SS) * sizeof(float);
  auto kern = acattn_bwd_fast_kernel<DH>;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(kern, dim3(p.B * p.n_heads), dim3(64 * nT), lds, stream, p, io);
  return (int)hipGetLastError();
}

}  // namespace

// Returns -100 when the problem is outside the fast path's domain (the caller then uses the general kernel).
int acattn_launch_bwd_fast(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  const bool ok = p.L <= 64 && (int64_t)p.B * p.n_heads * p.L * p.L < (1LL << 30) && (int64_t)p.B * p.L * p.H < (1LL << 30) &&
                  p.mask_mode == ACATTN_MASK_STRUCTURED && p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist &&
                  p.adversarial && p.combine_option == ACATTN_COMBINE_GATE && p.two_level;
  if (!ok) return -100;
  switch (p.H / p.n_heads) {
    case 16: return launch_fast<16>(p, io, stream);
    case 32: return launch_fast<32>(p, io, stream);
    case 64: return launch_fast<64>(p, io, stream);
  }
  return -100;
}

#ifdef ACATTN_BWD_STAMPS
extern "C" int acattn_debug_bwd_stamps(unsigned long long* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_stamps), (size_t)n_words * 8);
}
#endif
