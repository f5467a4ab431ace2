// Fused calibrated attention forward -- the training hot path (L <= 64), tuned for gfx950.
//
// Same mathematics and lane layout as acattn_fwd.hip (the general kernel, which stays the fallback for
// every other option combination) but with the shipped training configuration fixed at compile time:
// structured (item_seq != 0, causal | bidirectional) mask, in-kernel counter RNG, `gate` combine, both
// spatial calibrator terms, two_level.  What that buys:
//
//   * one wave per 16-row query block, 4 workgroups of 4 waves resident per CU (<= 128 VGPRs):
//     a wave's MFMA phases overlap the VALU phases of its neighbours on the same SIMD;
//   * K, Ka and V of the head are staged ONCE per workgroup into LDS with fully coalesced 16-byte loads
//     ("LDS-staged K/V tiles"); the key halves of the two spatial affines fall out of the staging pass;
//   * the block body is compiled per number of key tiles (1..4): no per-tile branches, so the loads of
//     a block are all in flight before its first MFMA;
//   * rows are streamed tile by tile through three passes that keep only P and one scratch tile set
//     live: pass 1 scores -> Pt, Mt;  pass 2 (dropout, M store, perturbed branch, exp of the calibrated
//     branch) feeds A_p straight into the P.V MFMAs;  pass 3 (gate, final softmax) feeds A_w likewise.
//     With bounded in-kernel noise no row maximum is needed beyond the first-level softmaxes: the
//     re-softmax inputs are probabilities, so the row's mask maximum is a safe shift;
//   * exponentials run in the exp2 domain (log2(e) folded into the scales), transcendentals are single
//     v_exp/v_log/v_rcp instructions.
#include <type_traits>

#include "acattn_common.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int DH, bool ADV>
__global__ void __launch_bounds__(256, 4) acattn_fwd_fast_kernel(const acattn_problem P, const acattn_fwd_out O) {
  constexpr int KS = DH / 4;
  constexpr int DT = DH / 16;
  constexpr int VS = DH + 4;
  constexpr int NT = 4;

  const int L = P.L, H = P.H, nh = P.n_heads;
  const int nT = (L + 15) >> 4;
  const int LP = nT * 16;
  int b, h;
  decode_block(blockIdx.x, P.B, nh, b, h);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> SGPRs for qb, i0, nt
  const int c = lane & 15, g = lane >> 4;
  const size_t rowbase = (size_t)b * L;
  const int hoff = h * DH;
  const size_t bh = (size_t)b * nh + h;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ks = smem;             // [LP][VS]
  float* Kas = Ks + LP * VS;    // [LP][VS]  (ADV only)
  float* Vs = Kas + (ADV ? LP * VS : 0);
  float* s_co = Vs + LP * VS;   // key half of the order affine
  float* s_cd = s_co + LP;      // key half of the distance affine
  float* s_km = s_cd + LP;      // key mask in the exp2 domain: 0 / -10000*log2e / -inf (j >= L)
  float* s_lt = s_km + LP;      // log(d + 1)
  const int GS = (L + 3) & ~3;  // row stride of the staged gate logits (pad columns are zero)
  float* Gs = s_lt + LP;        // [L][GS] gate logits of sequence b (ADV only)

  // ---- this wave's query block: fragments straight from HBM, issued before the staging barrier -------
  const int qb = wave, i0 = qb * 16, i = i0 + c;
  const bool row_ok = i < L;
  float qf[KS], qaf[KS];
  {
    const size_t off = (rowbase + (row_ok ? i : 0)) * H + hoff + KS * g;
#pragma unroll
    for (int s4 = 0; s4 < KS / 4; ++s4) {
      f4 t = {0.f, 0.f, 0.f, 0.f}, ta = t;
      if (row_ok) {
        t = *(const f4*)(P.q + off + 4 * s4);
        if (ADV) ta = *(const f4*)(P.qa + off + 4 * s4);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        qf[4 * s4 + e] = t[e];
        qaf[4 * s4 + e] = ta[e];
      }
    }
  }

  // ---- stage K, Ka, V (coalesced 16-byte loads) and derive the key-side calibrator terms --------------
  for (int idx = threadIdx.x; idx < LP * (DH / 4); idx += blockDim.x) {
    const int row = idx / (DH / 4), c4 = idx - row * (DH / 4);
    f4 kv = {0.f, 0.f, 0.f, 0.f}, kav = kv, vv = kv;
    if (row < L) {
      const size_t o = (rowbase + row) * H + hoff + 4 * c4;
      kv = *(const f4*)(P.k + o);
      if (ADV) kav = *(const f4*)(P.ka + o);
      vv = *(const f4*)(P.v + o);
    }
    *(f4*)(Ks + row * VS + 4 * c4) = kv;
    if (ADV) *(f4*)(Kas + row * VS + 4 * c4) = kav;
    *(f4*)(Vs + row * VS + 4 * c4) = vv;
    const f4 wo = *(const f4*)(P.w_order + DH + 4 * c4), wd = *(const f4*)(P.w_dist + DH + 4 * c4);
    float co = kv.x * wo.x + kv.y * wo.y + kv.z * wo.z + kv.w * wo.w;
    float cd = kv.x * wd.x + kv.y * wd.y + kv.z * wd.z + kv.w * wd.w;
#pragma unroll
    for (int off = 1; off < DH / 4; off <<= 1) {  // the DH/4 adjacent lanes of one key row
      co += __shfl_xor(co, off);
      cd += __shfl_xor(cd, off);
    }
    if (c4 == 0) {
      s_co[row] = co;
      s_cd[row] = cd;
    }
  }
  if (ADV) {
    // gate logits [L, L] of this sequence: one contiguous chunk, copied with coalesced 8-byte accesses;
    // inside the block body they are then LDS reads with no HBM latency (and no ordering behind M stores)
    const float* gsrc = P.gate_logits + rowbase * L;
    if ((L & 1) == 0) {
      const int half = GS >> 1, lhalf = L >> 1;
      for (int idx = threadIdx.x; idx < L * half; idx += blockDim.x) {
        const int row = idx / half, c2 = idx - row * half;
        float2 v = {0.f, 0.f};
        if (c2 < lhalf) v = *(const float2*)(gsrc + row * L + 2 * c2);
        *(float2*)(Gs + row * GS + 2 * c2) = v;
      }
    } else {
      for (int idx = threadIdx.x; idx < L * GS; idx += blockDim.x) {
        const int row = idx / GS, col = idx - row * GS;
        Gs[idx] = col < L ? gsrc[row * L + col] : 0.f;
      }
    }
  }
  if (threadIdx.x < LP) {
    const int j = threadIdx.x;
    float km = ACATTN_NEG_INF;
    if (j < L) km = P.key_valid[rowbase + j] ? 0.f : ACATTN_MASK_FILL * kLog2e;
    s_km[j] = km;
    s_lt[j] = logf((float)(j + 1));
  }
  // query halves of the two spatial affines (rank-1 form of layers.py:705-708,718,726)
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
  __syncthreads();
  if (qb >= nT) return;  // (block size is 64 * nT, so this never triggers; kept as a guard)

  const unsigned long long valid_keys = __ballot(lane < L && s_km[lane] == 0.f);
  const int first_valid = valid_keys ? __ffsll((long long)valid_keys) - 1 : L;
  const bool causal = P.causal != 0;
  // Key tiles that cannot receive probability mass are skipped: beyond the causal diagonal, and past the
  // last real item (right-padded sequences) -- legal only if every row of the block sees at least one
  // unmasked key (a fully masked row spreads over ALL keys, see acattn_fwd.hip).
  const int nt_valid = valid_keys ? ((63 - __clzll((long long)valid_keys)) >> 4) + 1 : nT;
  const bool rows_see_a_key = causal ? first_valid <= i0 : valid_keys != 0;
  const int nt = rows_see_a_key ? min(causal ? min(nT, qb + 1) : nT, nt_valid) : nT;

  const float hs2 = 0.5f * (sc * sc);
  const float inv_sqrt = 1.0f / sqrtf((float)DH);
  const float scale2 = inv_sqrt * kLog2e;  // scores -> exp2 domain
  const bool has_drop = P.p_drop > 0.f;
  const float keep_scale = has_drop ? 1.0f / (1.0f - P.p_drop) : 1.0f;
  const size_t prow = (bh * L + (row_ok ? i : 0)) * (size_t)L;
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

  auto body = [&](auto ntb_c) {
    constexpr int NTB = decltype(ntb_c)::value;

    // ---- pass 1: S^T = K.Q^T, Sa^T = Ka.Qa^T; spatial calibrator; first-level softmaxes ----------------
    f4 tS[NTB], tM[NTB];
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
      __builtin_amdgcn_sched_barrier(0);
      const float* kp = Ks + (16 * t + c) * VS + KS * g;
      const float* kap = Kas + (16 * t + c) * VS + KS * g;
      f4 aS = {0.f, 0.f, 0.f, 0.f}, aM = aS;
#pragma unroll
      for (int s4 = 0; s4 < KS / 4; ++s4) {
        const f4 k4 = *(const f4*)(kp + 4 * s4);
        f4 ka4;
        if (ADV) ka4 = *(const f4*)(kap + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          aS = mfma16(k4[e], qf[4 * s4 + e], aS);
          if (ADV) aM = mfma16(ka4[e], qaf[4 * s4 + e], aM);
        }
      }
      tS[t] = aS;
      tM[t] = aM;
    }
    float mx = ACATTN_NEG_INF, my = ACATTN_NEG_INF, shl = ACATTN_NEG_INF;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
      const f4 co4 = *(const f4*)(s_co + 16 * t + 4 * g);
      const f4 cd4 = *(const f4*)(s_cd + 16 * t + 4 * g);
      const f4 km4 = *(const f4*)(s_km + 16 * t + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * t + 4 * g + r;
        const bool fut = j > i;
        const float mkl = (causal && fut) ? fminf(km4[r], ACATTN_MASK_FILL * kLog2e) : km4[r];
        const float pr = fast_rcp(1.0f + ex2(-kLog2e * (ao + co4[r])));
        const float val = fut ? pr : 1.0f - pr;
        float s = tS[t][r] + __builtin_amdgcn_logf(val + ACATTN_LOG_EPS) * kLn2;  // layers.py:718-719
        const int dist = fut ? j - i : i - j;
        const float df = s_lt[dist] - (ad + cd4[r]);
        s -= (df * df) * hs2;  // layers.py:726-727
        const float x = s * scale2 + mkl;
        const float y = tM[t][r] * scale2 + mkl;
        tS[t][r] = x;
        tM[t][r] = y;
        mx = fmaxf(mx, x);
        my = fmaxf(my, y);
        shl = fmaxf(shl, mkl);
      }
    }
    mx = quad_max(mx);
    if (ADV) my = quad_max(my);
    shl = quad_max(shl);  // row's mask maximum (exp2 domain): 0, or -10000*log2e for a fully masked row
    float zx = 0.f, zy = 0.f;
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = ex2(tS[t][r] - mx);
        tS[t][r] = e;
        zx += e;
        if (ADV) {
          const float f = ex2(tM[t][r] - my);
          tM[t][r] = f;
          zy += f;
        }
      }
    }
    zx = quad_sum(zx);
    if (ADV) zy = quad_sum(zy);
    const float rzx = fast_rcp(zx), rzy = ADV ? fast_rcp(zy) : 0.f;

    f4 ca[DT], cc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      ca[dt] = f4{0.f, 0.f, 0.f, 0.f};
      cc[dt] = ca[dt];
    }
    float zu = 0.f, zv = 0.f, zw = 0.f;

    // ---- pass 2: dropout, M out, perturbed branch into P.V, exp of the calibrated branch ------------------
#pragma unroll
    for (int t = 0; t < NTB; ++t) {
      // keep one tile's working set live at a time: without the fence the scheduler hoists the RNG and
      // LDS reads of all tiles to the top and the kernel no longer fits 4 waves per SIMD
      __builtin_amdgcn_sched_barrier(0);
      const f4 km4 = *(const f4*)(s_km + 16 * t + 4 * g);
      uint32_t ka = 0xFu, km_ = 0xFu;
      float nz[4] = {0.f, 0.f, 0.f, 0.f};
      if (ADV || has_drop) {
        const RngGroup rg = rng_group(P.seed, rng_row, (uint32_t)(4 * t + g), P.p_drop);
#pragma unroll
        for (int r = 0; r < 4; ++r) nz[r] = rg.n[r];
        if (has_drop) {
          ka = rg.keep_after;
          km_ = rg.keep_mask;
        }
      }
      f4 eu;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * t + 4 * g + r;
        const float mkl = (causal && j > i) ? fminf(km4[r], ACATTN_MASK_FILL * kLog2e) : km4[r];
        const float p = ((ka >> r) & 1u) ? tS[t][r] * (rzx * keep_scale) : 0.f;  // P   layers.py:735-736
        tS[t][r] = p;
        if (ADV) {
          const float m = ((km_ >> r) & 1u) ? tM[t][r] * (rzy * keep_scale) : 0.f;  // M   layers.py:670-672
          tM[t][r] = m;
          const float u = p * m + nz[r] * (1.0f - m);  // layers.py:918
          eu[r] = ex2(u * kLog2e + (mkl - shl));
          zu += eu[r];
        }
      }
      if (ADV) {
        store_seg(O.attack_mask, t, tM[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = 16 * t + 4 * g + r;
          const float mkl = (causal && j > i) ? fminf(km4[r], ACATTN_MASK_FILL * kLog2e) : km4[r];
          const float v_ = tS[t][r] * ex2(kLog2e - tM[t][r] * kLog2e);  // P * exp(1 - M)   layers.py:920
          const float e = ex2(v_ * kLog2e + (mkl - shl));
          tM[t][r] = e;  // M is dead from here on: keep the unnormalised A_c in its registers
          zv += e;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* vp = Vs + (16 * t + 4 * g + r) * VS + c;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          if (ADV)
            ca[dt] = mfma16(vp[16 * dt], eu[r], ca[dt]);
          else
            cc[dt] = mfma16(vp[16 * dt], tS[t][r], cc[dt]);  // spatial calibrator only: ctx = P.V
        }
      }
    }

    // ---- pass 3: gate combine, final softmax, calibrated branch into P.V ---------------------------------
    if (ADV) {
      zv = quad_sum(zv);
      const float rzv = fast_rcp(zv);
#pragma unroll
      for (int t = 0; t < NTB; ++t) {
        __builtin_amdgcn_sched_barrier(0);
        const f4 km4 = *(const f4*)(s_km + 16 * t + 4 * g);
        const int j0 = 16 * t + 4 * g;
        // staged logits; key groups past the row end read a clamped (finite) address, their A_g is 0 anyway
        const f4 gl = *(const f4*)(Gs + (row_ok ? i : 0) * GS + min(j0, GS - 4));
        f4 ew;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + r;
          const float mkl = (causal && j > i) ? fminf(km4[r], ACATTN_MASK_FILL * kLog2e) : km4[r];
          const float gt = fast_rcp(1.0f + ex2(-kLog2e * gl[r]));
          const float acn = tM[t][r] * rzv;                      // A_c
          const float ag = gt * (tS[t][r] - acn) + acn;          // layers.py:888
          ew[r] = ex2(ag * kLog2e + (mkl - shl));                // layers.py:925
          zw += ew[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float* vp = Vs + (16 * t + 4 * g + r) * VS + c;
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) cc[dt] = mfma16(vp[16 * dt], ew[r], cc[dt]);
        }
      }
      zu = quad_sum(zu);
      zw = quad_sum(zw);
      const float rzu = fast_rcp(zu), rzw = fast_rcp(zw);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        ca[dt] *= rzu;
        cc[dt] *= rzw;
      }
      for (int t = NTB; t < nT; ++t) store_seg(O.attack_mask, t, f4{0.f, 0.f, 0.f, 0.f});  // causally skipped tiles
    }

    if (row_ok) {
      float* oc = O.ctx_calibrated + (rowbase + i) * H + hoff + 4 * g;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) *(f4*)(oc + 16 * dt) = cc[dt];
      if (ADV) {
        float* oa = O.ctx_attacked + (rowbase + i) * H + hoff + 4 * g;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) *(f4*)(oa + 16 * dt) = ca[dt];
        if (O.row_stats && g == 0) {
          // natural-log normalisers, as the backward expects them (acattn_bwd.hip)
          float* sp = O.row_stats + (bh * L + i) * ACATTN_NSTAT;
          const float sh = shl * kLn2;
          *(f4*)sp = f4{(mx + __builtin_amdgcn_logf(zx)) * kLn2, (my + __builtin_amdgcn_logf(zy)) * kLn2,
                        sh + fast_log(zu), sh + fast_log(zv)};
          *(f4*)(sp + 4) = f4{sh + fast_log(zw), 0.f, 0.f, 0.f};
        }
      }
    }
  };

  switch (nt) {
    case 1: body(std::integral_constant<int, 1>{}); break;
    case 2: body(std::integral_constant<int, 2>{}); break;
    case 3: body(std::integral_constant<int, 3>{}); break;
    default: body(std::integral_constant<int, 4>{}); break;
  }
}

template <int DH>
int launch_fast(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  const int nT = (p.L + 15) / 16, LP = nT * 16;
  const int GS = (p.L + 3) & ~3;
  const size_t lds = (size_t)((p.adversarial ? 3 : 2) * LP * (DH + 4) + 4 * LP + (p.adversarial ? p.L * GS : 0)) * sizeof(float);
  const dim3 grid(p.B * p.n_heads), block(64 * nT);
  if (p.adversarial)
    hipLaunchKernelGGL((acattn_fwd_fast_kernel<DH, true>), grid, block, lds, stream, p, o);
  else
    hipLaunchKernelGGL((acattn_fwd_fast_kernel<DH, false>), grid, block, lds, stream, p, o);
  return (int)hipGetLastError();
}

}  // namespace

// Returns -100 when the problem is outside the fast path's domain (the caller then uses the general kernel).
int acattn_launch_fwd_fast(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  const bool ok = p.L <= 64 && p.mask_mode == ACATTN_MASK_STRUCTURED && p.rng_mode == ACATTN_RNG_COUNTER && p.w_order &&
                  p.w_dist && (!p.adversarial || (p.combine_option == ACATTN_COMBINE_GATE && p.two_level)) &&
                  !o.after_spatial && !o.before_spatial && !o.perturbed_attention && !o.calibrated_attention;
  if (!ok) return -100;
  switch (p.H / p.n_heads) {
    case 16: return launch_fast<16>(p, o, stream);
    case 32: return launch_fast<32>(p, o, stream);
    case 64: return launch_fast<64>(p, o, stream);
  }
  return -100;
}
