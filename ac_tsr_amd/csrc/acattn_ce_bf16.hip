// Full-catalogue cross-entropy, hidden 64, with the three products on the bf16 matrix pipe at fp32 accuracy [round 4].
//
// Same algorithm, decomposition and outputs as acattn_ce.hip (items stationary, a wave owns 16*TILES table rows and
// sweeps the batch; ACSASRec._cal_loss, recbole/model/sequential_recommender/acsasrec.py:117-120).  What changes is the
// arithmetic of the products.  gfx950 has no fast fp32 matrix instruction: v_mfma_f32_16x16x4_f32 runs at the vector
// rate, 1/16 of v_mfma_f32_16x16x32_bf16.  Every fp32 operand x is therefore split EXACTLY into three bf16 numbers,
//     x = x0 + x1 + x2,   x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1)      (3 x 8 = 24 significand bits)
// and a product a.b is evaluated as the six bf16 MFMAs with i + j <= 2,
//     a.b ~ a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0),
// each exact in the fp32 accumulator (8 x 8 bit significands); what is dropped (a1 b2 + a2 b1 + a2 b2) is below
// 2^-23 |a||b|, the size of ONE fp32 rounding of the product.  Six 16-cycle MFMAs of K = 32 replace eight 32-cycle
// MFMAs of K = 4: 0.375 of the matrix time.  Measured against fp64 the results are as close as the exact-fp32 kernels'
// (tests/test_hip_ce.py::test_split_products_are_as_accurate_as_fp32_products).
//
// Layouts (lane = 16 g + c).  v_mfma_f32_16x16x32_bf16: A[m = c][k = 8g + j], B[k = 8g + j][n = c], j < 8;
// D[m = 4g + r][n = c].
//   P1  logits^T[item, row] = E . out^T        A = Er[t][s][p]  (item 16t + c, channels 32s + 8g + j), B = rows (row c,
//       same channels).  D: items 16t + 4g + r in registers, batch row c on the lane -- as in acattn_ce.hip, so the
//       soft-max code is that file's.
//   P2  d out^T[ch, row]   = E^T . dl^T        B = the dl registers as they stand: k = 8g + 4tt + r is item
//       16(2u + tt) + 4g + r of tile pair u;  A = Ec[u][cb][p] holds the table in that item order (channel 16cb + c).
//   P3  d E[item, ch]     += dl^T . out        sums over batch rows, which sit on the lane: dl goes through a per-wave
//       LDS image [batch row][item] (8-byte stores of 4 items) and comes back transposed with ds_read_b64_tr_b16
//       (A[m = item c][k = row 8g + j]); B = rows by column (channel 16cb + c, rows 8g + j).  K = 32 rows: the sweep
//       advances in SUPER-BLOCKS of 32 batch rows, P1 / soft-max / P2 per 16-row half.
// The table operands (both orders, three planes each) stay in registers for the whole sweep: 288 of the 512 a wave has
// at one wave per SIMD, the second order pinned in accumulator registers (MFMAs read A / B from there directly; left to the
// allocator the excess over 256 is spilled there and copied back in front of every use).  The batch rows are split once
// per launch by ce_split_rows_kernel into the exact operand images (24 slots of 64 lanes x 16 bytes per super-block) and
// reach LDS by LDS-DMA into a double buffer, one super-block ahead.  The target's one-hot is applied behind the sweep
// (ce6_onehot_reduce_kernel); the catalogue's leftover tiles behind whole rounds of workgroups run on the same code with
// one tile.  What was measured on the way: DESIGN.md 4.5, profiles/r04_ce6_stamps.txt.
#include <stdlib.h>

#include "acattn_common.h"

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr int CH = 64;
constexpr int NW = 4;            // waves per workgroup
constexpr int ES = CH + 4;       // fp32 row stride of a parked [rows][CH] tile
constexpr int HSLOTS = 24;       // operand slots of a super-block: 12 by row ((h, s, q)), 12 by column ((q, cb))
constexpr int HB_BYTES = HSLOTS * 64 * 16;
#ifndef CE6_EC_AGPR
#define CE6_EC_AGPR 1
#endif
#ifndef CE6_PIPE
#define CE6_PIPE 0  // vector instructions asked for behind every MFMA of a pipelined product; 0 = phases in program order
#endif

typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef __bf16 b4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));

// product terms (plane of the first operand, plane of the second), smallest first
#define CE6_TERMS(X) X(0, 2) X(1, 1) X(2, 0) X(0, 1) X(1, 0) X(0, 0)

__device__ __forceinline__ f4 mfma_bf(const b8 a, const b8 b, const f4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

// One wave-wide LDS-DMA: lane l copies the 16 bytes at gsrc (per lane) to LDS byte address lds_dst + 16 l (lds_dst
// wave-uniform, in M0) without passing through registers.  Inline assembly as in acattn_fwd_dma.hip: the kernel orders
// the copies itself (s_waitcnt vmcnt(0) in front of the barrier that publishes them).
__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  const uint32_t lds_off = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_off) : "memory", "m0");
}

// x = p0 + p1 + p2 exactly (round-to-nearest pieces; v_cvt_pk_bf16_f32)
__device__ __forceinline__ void split8(const float (&x)[8], b8& p0, b8& p1, b8& p2) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 a = (__bf16)x[j];
    const float r1 = x[j] - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    p0[j] = a;
    p1[j] = b;
    p2[j] = (__bf16)r2;
  }
}

// One workgroup per super-block of 32 batch rows: the rows' operand images.
//   slot (2h + s) * 3 + q,  lane (c, g): plane q of out[32 sb + 16 h + c][32 s + 8 g + j]          (P1's B operand)
//   slot 12 + 4 q + cb,     lane (c, g): plane q of out[32 sb + 8 g + j][16 cb + c]                  (P3's B operand)
// Also zeroes `n_zero` floats at `zero` (the leftover tiles' d_table rows when several workgroups add into them).
__global__ void __launch_bounds__(256) ce_split_rows_kernel(const float* __restrict__ out, const int B, b8* __restrict__ Hb,
                                                            float* __restrict__ zero, const int64_t n_zero) {
  const int sb = blockIdx.x;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_zero; i += (int64_t)gridDim.x * 256) zero[i] = 0.f;
  for (int e = threadIdx.x; e < 8 * 64; e += 256) {
    const int grp = e >> 6, lane = e & 63, c = lane & 15, g = lane >> 4;
    float x[8];
    if (grp < 4) {
      const int row = 32 * sb + 16 * (grp >> 1) + c;
      f4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
      if (row < B) {
        v0 = *(const f4*)(out + (size_t)row * CH + 32 * (grp & 1) + 8 * g);
        v1 = *(const f4*)(out + (size_t)row * CH + 32 * (grp & 1) + 8 * g + 4);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = v0[j];
        x[4 + j] = v1[j];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int row = 32 * sb + 8 * g + j;
        x[j] = row < B ? out[(size_t)row * CH + 16 * (grp - 4) + c] : 0.f;
      }
    }
    b8 p[3];
    split8(x, p[0], p[1], p[2]);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int slot = grp < 4 ? grp * 3 + q : 12 + 4 * q + (grp - 4);
      Hb[((size_t)sb * HSLOTS + slot) * 64 + lane] = p[q];
    }
  }
}

// Diagnostic builds only (-DACATTN_CE_STAMPS, tools/gpu_ce6_stamps.sh): cycles per phase, summed per wave.
#ifdef ACATTN_CE_STAMPS
__device__ unsigned long long g_ce6_stamps[4096 * 8];
#define CE6_STAMP(k)                                                               \
  do {                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                             \
    unsigned long long now_;                                                       \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");   \
    if ((k) >= 0) cyc_[(k) >= 0 ? (k) : 0] += now_ - last_;                        \
    last_ = now_;                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                             \
  } while (0)
#else
#define CE6_STAMP(k)
#endif
// -DCE6_STAMP_LEFT (with ACATTN_CE_STAMPS): slots 0..5 are re-used for the phases of the leftover units instead
#if defined(ACATTN_CE_STAMPS) && defined(CE6_STAMP_LEFT)
#define CE6_LSTAMP(k) CE6_STAMP(k)
#define CE6_LRESET() do { for (int z_ = 0; z_ < 6; ++z_) cyc_[z_] = 0; } while (0)
#else
#define CE6_LSTAMP(k)
#define CE6_LRESET()
#endif

template <int TILES>
struct Ce6 {
  static_assert(TILES % 2 == 0, "P2 takes the item tiles in pairs (K = 32)");
  static constexpr int ITEMS = 16 * TILES;
  static constexpr int UP = TILES / 2;
  static constexpr int XRS = 2 * ITEMS + 8;            // bytes per batch row of the transpose image (8-byte aligned, 50 dwords at 6 tiles)
  static constexpr int XPLANE = 32 * XRS;
  static constexpr int XBYTES_RAW = 3 * XPLANE;
  // the leftover units' working set (acattn_ce.hip's: a tile's rows, a row block, a transpose scratch, fp32) shares the area
  static constexpr int LEFT_BYTES = (32 * ES + 16 * 48) * 4;
  static constexpr int XBYTES = ((XBYTES_RAW > LEFT_BYTES ? XBYTES_RAW : LEFT_BYTES) + 15) / 16 * 16;
  static constexpr int PARK_FLOATS = 32 * ES;          // per wave
  static constexpr int LDS_BYTES = 2 * HB_BYTES + NW * XBYTES + NW * PARK_FLOATS * 4;
  static_assert(LDS_BYTES <= 160 * 1024, "one workgroup per CU");
};

// DIR as in acattn_ce.hip: a forward that also yields the direction of d_out (running maxima per wave, folded per
// workgroup, finished by ce_dir_reduce_kernel).
template <int TILES, bool WITH_TABLE_GRAD, bool DIR>
__global__ void __launch_bounds__(64 * NW) ce6_bwd_kernel(const acattn_ce_problem P, const float* __restrict__ lse,
                                                          const float* __restrict__ coef, float* __restrict__ d_out_slab,
                                                          float* __restrict__ d_table, float2* __restrict__ part,
                                                          const b8* __restrict__ Hb, const int n_left) {
  static_assert(!(DIR && WITH_TABLE_GRAD), "the forward-with-direction sweep has no table gradient");
  using C = Ce6<TILES>;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int item0 = (blockIdx.x * NW + wave) * C::ITEMS;
  const int B = P.B, N = P.N;
  const bool ragged = item0 + C::ITEMS > N;  // (uniform per wave)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  // [2][24][64] operand images of the current and the next super-block (filled by LDS-DMA one super-block ahead)
  char* Xw = smem + 2 * HB_BYTES + wave * C::XBYTES;            // this wave's transpose image, 3 planes [32 rows][XRS]
  float* park = (float*)(smem + 2 * HB_BYTES + NW * C::XBYTES); // [NW][32][ES] parked d_out tiles
  float* Pw = park + wave * C::PARK_FLOATS;

#ifdef ACATTN_CE_STAMPS
  unsigned long long cyc_[8] = {}, last_ = 0;
#endif
  CE6_STAMP(-1);
  // ---- the wave's table rows, split, in both operand orders -------------------------------------------------------
  b8 Er[TILES][2][3];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
    const int item = item0 + 16 * t + c;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // (clamped address + select: a branch per load would serialise the prologue's 100+ requests)
      const float* src = P.table + (size_t)min(item, N - 1) * CH + 32 * s + 8 * g;
      const f4 v0 = *(const f4*)src, v1 = *(const f4*)(src + 4);
      float x[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = item < N ? v0[j] : 0.f;
        x[4 + j] = item < N ? v1[j] : 0.f;
      }
      split8(x, Er[t][s][0], Er[t][s][1], Er[t][s][2]);
    }
  }
  b8 Ec[C::UP][4][3];
#pragma unroll
  for (int u = 0; u < C::UP; ++u) {
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int item = item0 + 16 * (2 * u + (j >> 2)) + 4 * g + (j & 3);
        const float v = P.table[(size_t)min(item, N - 1) * CH + 16 * cb + c];
        x[j] = item < N ? v : 0.f;
      }
      split8(x, Ec[u][cb][0], Ec[u][cb][1], Ec[u][cb][2]);
#if CE6_EC_AGPR
      // The two table images are 288 registers, more than the 256 architectural ones: left to itself the allocator parks the
      // excess in accumulator registers as SPILLS and copies four dwords back in front of every MFMA that uses them.  An
      // MFMA reads its A / B operands from accumulator registers directly: this image is pinned there for the sweep.
#pragma unroll
      for (int p = 0; p < 3; ++p) asm volatile("" : "+a"(Ec[u][cb][p]));
#endif
    }
  }
  f4 dE[TILES][4];
#pragma unroll
  for (int t = 0; t < TILES; ++t)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) dE[t][cb] = f4{0.f, 0.f, 0.f, 0.f};

  const int nsb = (B + 31) >> 5;
  // A super-block's images go global -> LDS by DMA, each wave a quarter (six 1-KB pieces), one super-block ahead; the
  // rows' lse / coef / target are fetched into registers one super-block ahead.
  auto dma_rows = [&](int sb) {
    const char* src = (const char*)Hb + (size_t)sb * HB_BYTES + (6 * wave) * 1024 + lane * 16;
    char* dst = smem + (sb & 1) * HB_BYTES + (6 * wave) * 1024;
#pragma unroll
    for (int u = 0; u < 6; ++u) dma16(src + u * 1024, dst + u * 1024);
  };
  float lse_next[2] = {0.f, 0.f}, cf_next[2] = {0.f, 0.f};
  auto prefetch = [&](int sb) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = 32 * sb + 16 * h + c;
      const bool ok = row < B;
      lse_next[h] = (ok && !DIR) ? lse[row] : 0.f;
      cf_next[h] = (ok && !DIR) ? coef[P.coef_is_scalar ? 0 : row] * (P.coef_scale != 0.f ? P.coef_scale : 1.0f) : 0.f;
    }
  };
  dma_rows(0);
  prefetch(0);
  float l2_cur[2], cf_cur[2];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  CE6_STAMP(6);
  for (int sb = 0; sb < nsb; ++sb) {
    const b8* Hs = (const b8*)(smem + (sb & 1) * HB_BYTES);
    if (sb + 1 < nsb) dma_rows(sb + 1);  // its buffer was last read before the previous super-block's first barrier
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      l2_cur[h] = lse_next[h] * kLog2e;
      cf_cur[h] = cf_next[h];
    }
    if (sb + 1 < nsb) prefetch(sb + 1);
    // The half's phases as pieces, so that they can be issued in a software-pipelined order below.
    auto p1 = [&](const int h, f4 (&dl)[TILES]) {  // P1: logits^T of the wave's items for the half's 16 rows
      b8 Hr[2][3];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q) Hr[s][q] = Hs[((2 * h + s) * 3 + q) * 64 + lane];
#pragma unroll
      for (int t = 0; t < TILES; ++t) dl[t] = f4{0.f, 0.f, 0.f, 0.f};
#define CE6_P1(p, q)                                                                  \
  _Pragma("unroll") for (int s = 0; s < 2; ++s)                                      \
      _Pragma("unroll") for (int t = 0; t < TILES; ++t) dl[t] = mfma_bf(Er[t][s][p], Hr[s][q], dl[t]);
      CE6_TERMS(CE6_P1)
#undef CE6_P1
    };
    auto ragged_fix = [&](f4 (&dl)[TILES]) {
      if (ragged) {
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (item0 + 16 * t + 4 * g + r >= N) dl[t][r] = ACATTN_NEG_INF;  // exp2(-inf) = 0 past the catalogue end
      }
    };
    // soft-max arithmetic (acattn_ce.hip's, the layout is the same), then the split of dl: its planes are P2's B operand
    // as they stand, and go to the transpose image for P3
    auto valu = [&](const int h, f4 (&dl)[TILES], b8 (&dlB)[C::UP][3], float& m_w, float& s_w) {
      if (DIR) {
        m_w = ACATTN_NEG_INF;
#pragma unroll
        for (int t = 0; t < TILES; ++t) m_w = fmaxf(fmaxf(fmaxf(fmaxf(m_w, dl[t][0]), dl[t][1]), dl[t][2]), dl[t][3]);
        m_w = quad_max(m_w);
        const float m2 = m_w > ACATTN_NEG_INF ? m_w * kLog2e : 0.f;
        f4 sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          f4 x = dl[t] * kLog2e - m2;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
          dl[t] = x;
          sv += x;
        }
        s_w = quad_sum((sv[0] + sv[1]) + (sv[2] + sv[3]));
      } else {
        // dl = coef * softmax.  The target's one-hot is not subtracted here: its two contributions, -coef E_target to the
        // row's d_out and -coef out_row to d_table[target], are added by ce6_onehot_reduce_kernel behind the sweep.
        const float l2 = l2_cur[h], cf = cf_cur[h];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          f4 x = dl[t] * kLog2e - l2;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
          dl[t] = x * cf;
        }
      }
#pragma unroll
      for (int u = 0; u < C::UP; ++u) {
        float x[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x[r] = dl[2 * u][r];
          x[4 + r] = dl[2 * u + 1][r];
        }
        split8(x, dlB[u][0], dlB[u][1], dlB[u][2]);
      }
      if (WITH_TABLE_GRAD) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int u = 0; u < C::UP; ++u) {
            char* dst = Xw + p * C::XPLANE + (16 * h + c) * C::XRS + (32 * u + 4 * g) * 2;
            *(b4*)dst = __builtin_shufflevector(dlB[u][p], dlB[u][p], 0, 1, 2, 3);         // items 16(2u) + 4g ..
            *(b4*)(dst + 32) = __builtin_shufflevector(dlB[u][p], dlB[u][p], 4, 5, 6, 7);  // items 16(2u + 1) + 4g ..
          }
      }
    };
    auto p2 = [&](const b8 (&dlB)[C::UP][3], f4 (&dh)[4]) {  // P2: d out^T (this wave's items) = E^T . dl^T
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) dh[cb] = f4{0.f, 0.f, 0.f, 0.f};
#define CE6_P2(p, q)                                                                  \
  _Pragma("unroll") for (int u = 0; u < C::UP; ++u)                                  \
      _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) dh[cb] = mfma_bf(Ec[u][cb][p], dlB[u][q], dh[cb]);
      CE6_TERMS(CE6_P2)
#undef CE6_P2
    };
    auto park_tile = [&](const int h, const f4 (&dh)[4], const float m_w, const float s_w) {  // the half's [16][CH] tile, for the fold
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) *(f4*)(Pw + (16 * h + c) * ES + 16 * cb + 4 * g) = dh[cb];
      if (DIR && g == 0) {  // (max, sum-exp) of this wave for the row, in the pad columns of its parked tile
        Pw[(16 * h + c) * ES + CH] = m_w;
        Pw[(16 * h + c) * ES + CH + 1] = s_w;
      }
    };
    // one MFMA, then `nv` vector instructions, `n` times: the order the scheduler is asked for inside the current region
    auto interleave = [&](auto n_mfma, auto n_valu) {
#pragma unroll
      for (int i = 0; i < decltype(n_mfma)::value; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, decltype(n_valu)::value, 0);
      }
    };
    float m_w[2] = {ACATTN_NEG_INF, ACATTN_NEG_INF}, s_w[2] = {0.f, 0.f};
    f4 dh[4];
#if CE6_PIPE
    // Software pipeline: the second half's logits product is issued with the first half's soft-max / split arithmetic
    // in between its MFMAs (a bf16 MFMA holds the issue port for half of its 16 cycles), and the first half's d_out
    // product with the second half's arithmetic.  One wave per SIMD: nothing else would fill those slots.
    CE6_STAMP(-1);
    f4 dl0[TILES], dl1[TILES];
    b8 dlB0[C::UP][3], dlB1[C::UP][3];
    p1(0, dl0);
    ragged_fix(dl0);
    CE6_STAMP(0);
    p1(1, dl1);
    valu(0, dl0, dlB0, m_w[0], s_w[0]);
    interleave(std::integral_constant<int, 12 * TILES>{}, std::integral_constant<int, CE6_PIPE>{});
    ragged_fix(dl1);
    CE6_STAMP(1);
    p2(dlB0, dh);
    valu(1, dl1, dlB1, m_w[1], s_w[1]);
    interleave(std::integral_constant<int, 24 * C::UP>{}, std::integral_constant<int, CE6_PIPE>{});
    CE6_STAMP(2);
    park_tile(0, dh, m_w[0], s_w[0]);
    p2(dlB1, dh);
    park_tile(1, dh, m_w[1], s_w[1]);
    CE6_STAMP(3);
#else
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      CE6_STAMP(-1);
      f4 dl[TILES];
      b8 dlB[C::UP][3];
      p1(h, dl);
      CE6_STAMP(0);
      ragged_fix(dl);
      valu(h, dl, dlB, m_w[h], s_w[h]);
      CE6_STAMP(2);
      p2(dlB, dh);
      park_tile(h, dh, m_w[h], s_w[h]);
      CE6_STAMP(3);
    }
#endif
    if (WITH_TABLE_GRAD) {
      // ---- P3: d E (this wave's items) += dl^T . out over the super-block's 32 rows --------------------------------
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the image is the wave's own: its stores have landed, no barrier
      b8 Hc[3][4];
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) Hc[q][cb] = Hs[(12 + 4 * q + cb) * 64 + lane];
#pragma unroll
      for (int t = 0; t < TILES; ++t) {
        b8 At[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          // lane 4q' + p' of a 16-lane group addresses row q', columns 4p' .. of the block; receives column c, rows 0..3
          const char* src = Xw + p * C::XPLANE + (8 * g + (c >> 2)) * C::XRS + (16 * t + 4 * (c & 3)) * 2;
          const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(src));
          const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(src + 4 * C::XRS));
          const s8v both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          At[p] = __builtin_bit_cast(b8, both);
        }
#define CE6_P3(p, q) \
  _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) dE[t][cb] = mfma_bf(At[p], Hc[q][cb], dE[t][cb]);
        CE6_TERMS(CE6_P3)
#undef CE6_P3
      }
    }
    CE6_STAMP(4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next super-block's images has landed
    __syncthreads();  // every wave's tiles are parked, the next images are complete
    // the fold of this super-block's d_out tiles (each thread sums float4s of the four waves)
    for (int idx = threadIdx.x; idx < 32 * (CH / 4); idx += 64 * NW) {
      const int r = idx / (CH / 4), c4 = idx - r * (CH / 4);
      if (32 * sb + r < B) {
        f4 sum = {0.f, 0.f, 0.f, 0.f};
        const size_t o = (size_t)(32 * sb + r) * CH + 4 * c4;
        if (DIR) {
          float mw[NW], m_wg = ACATTN_NEG_INF;
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            mw[w] = park[w * C::PARK_FLOATS + r * ES + CH];
            m_wg = fmaxf(m_wg, mw[w]);
          }
          float s_wg = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) {
            const float sc = mw[w] > ACATTN_NEG_INF ? __builtin_amdgcn_exp2f((mw[w] - m_wg) * kLog2e) : 0.f;
            sum += *(const f4*)(park + w * C::PARK_FLOATS + r * ES + 4 * c4) * sc;
            s_wg += park[w * C::PARK_FLOATS + r * ES + CH + 1] * sc;
          }
          *(f4*)(d_out_slab + (size_t)blockIdx.x * B * CH + o) = sum;
          if (c4 == 0) part[(size_t)blockIdx.x * B + 32 * sb + r] = float2{m_wg, s_wg};
          continue;
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += *(const f4*)(park + w * C::PARK_FLOATS + r * ES + 4 * c4);
        *(f4*)(d_out_slab + (size_t)blockIdx.x * B * CH + o) = sum;
      }
    }
    __syncthreads();  // the parked tiles are free
    CE6_STAMP(5);
  }
  // The wave's d_table rows.  A lane holds, per tile, 16 single floats of four different rows (item 4g + r, channel
  // 16cb + c): stored as they stand that is 96 dword stores of four 64-byte pieces each -- 12,500 cycles per wave,
  // 6 us of the launch (tools/gpu_ce6_stamps.sh, CE6_STAMP_LEFT=1).  Through the wave's parked-tile area instead: a tile
  // as [16 items][64 channels], read back as whole rows, four 16-byte stores per lane and tile (256 contiguous bytes
  // per 16 lanes).
  auto store_dE = [&]() {
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pw[(4 * g + r) * ES + 16 * cb + c] = dE[t][cb][r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the area is the wave's own)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int row = g + 4 * k, item = item0 + 16 * t + row;
        const f4 v = *(const f4*)(Pw + row * ES + 4 * c);
        if (item < N) *(f4*)(d_table + (size_t)item * CH + 4 * c) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the reads are done before the next tile overwrites the area
    }
  };
  if (WITH_TABLE_GRAD && n_left <= 0) store_dE();
  if (n_left > 0) {
    // Leftover tiles behind the whole rounds (acattn_ce.hip's scheme): a UNIT is (leftover tile, super-block).  With a
    // table gradient workgroup l < n_left takes all of tile l, its waves every fourth super-block, so that the tile's
    // d_table rows are summed inside the workgroup; without one the units are dealt out evenly over all waves.  Same
    // three products on one item tile (P2's K = 32 half empty); every wave works on its own, its operand images come
    // straight from global memory (L2) one unit ahead -- the table operands' registers are free now.
    // Results: slab / partial number gridDim.x + tile.
    const int wid = blockIdx.x * NW + wave;
    int first, stride, n_my;
    // (with a table gradient: `pieces` workgroups share a tile, a run of super-blocks each, when there are at least twice
    // as many workgroups as leftover tiles; their d_table sums are then ADDED to rows that ce_split_rows_kernel zeroed)
    const int pieces = WITH_TABLE_GRAD ? max(1, (int)gridDim.x / n_left) : 1;
    const int tile_mine = blockIdx.x / pieces;
    if (WITH_TABLE_GRAD) {
      const int per = (nsb + pieces - 1) / pieces, lo = (blockIdx.x % pieces) * per, hi = min(nsb, lo + per);
      first = tile_mine * nsb + lo + wave;
      stride = NW;
      n_my = tile_mine < n_left ? max(0, (hi - lo - wave + NW - 1) / NW) : 0;
    } else {
      const int n_units = n_left * nsb, U = (n_units + gridDim.x * NW - 1) / (gridDim.x * NW);
      first = wid * U;
      stride = 1;
      n_my = min(max(n_units - first, 0), U);
    }
    constexpr int XRS1 = 40, XPL1 = 32 * XRS1;  // one-tile transpose image: 16 items + pad per batch row
    b8 Er1[2][3], Ec1[4][3];
    f4 dE1[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) dE1[cb] = f4{0.f, 0.f, 0.f, 0.f};
    // A unit's operands (24 images, the rows' lse / coef; on a tile change the tile's table rows) are REQUESTED one unit
    // ahead -- the first unit's in front of the store of the main sweep's d_table rows -- and taken over when it starts:
    // a wave has two units or so and each request is a full L2 / HBM round trip.
    b8 Hn[HSLOTS];
    f4 er_n[2][2];
    float ec_n[4][4];
    float lse_n[2] = {0.f, 0.f}, cf_n[2] = {0.f, 0.f};
    auto request = [&](const int u, const bool with_tile) {
      const int tile = u / nsb, sb = u - tile * nsb;
      const b8* src = Hb + (size_t)sb * HSLOTS * 64 + lane;
#pragma unroll
      for (int i = 0; i < HSLOTS; ++i) Hn[i] = src[i * 64];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = 32 * sb + 16 * h + c;
        const bool ok = row < B;
        lse_n[h] = (ok && !DIR) ? lse[row] : 0.f;
        cf_n[h] = (ok && !DIR) ? coef[P.coef_is_scalar ? 0 : row] * (P.coef_scale != 0.f ? P.coef_scale : 1.0f) : 0.f;
      }
      if (with_tile) {  // (uniform per wave)
        const int itx = gridDim.x * NW * C::ITEMS + 16 * tile;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float* tsrc = P.table + (size_t)min(itx + c, N - 1) * CH + 32 * s + 8 * g;
          er_n[s][0] = *(const f4*)tsrc;
          er_n[s][1] = *(const f4*)(tsrc + 4);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
          for (int j = 0; j < 4; ++j) ec_n[cb][j] = P.table[(size_t)min(itx + 4 * g + j, N - 1) * CH + 16 * cb + c];
      }
    };
    CE6_LRESET();
    CE6_LSTAMP(-1);
    if (n_my > 0) request(first, true);
    if (WITH_TABLE_GRAD) store_dE();
    CE6_LSTAMP(0);
    int cur_tile = -1;
    for (int k = 0; k < n_my; ++k) {
      CE6_LSTAMP(-1);
      const int u = first + k * stride, tile = u / nsb, sb = u - tile * nsb;
      const int itx = gridDim.x * NW * C::ITEMS + 16 * tile;  // first item of the leftover tile
      const size_t vslab = (size_t)(gridDim.x + tile) * B;    // row offset of this tile's slab / partials
      if (tile != cur_tile) {  // (uniform per wave) the requested table rows -> split operands
        cur_tile = tile;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            x[j] = itx + c < N ? er_n[s][0][j] : 0.f;
            x[4 + j] = itx + c < N ? er_n[s][1][j] : 0.f;
          }
          split8(x, Er1[s][0], Er1[s][1], Er1[s][2]);
        }
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
          float x[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            x[j] = itx + 4 * g + j < N ? ec_n[cb][j] : 0.f;
            x[4 + j] = 0.f;
          }
          split8(x, Ec1[cb][0], Ec1[cb][1], Ec1[cb][2]);
        }
      }
      CE6_LSTAMP(1);
      b8 Hq[HSLOTS];
#pragma unroll
      for (int i = 0; i < HSLOTS; ++i) Hq[i] = Hn[i];
      float l2_u[2], cf_u[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        l2_u[h] = lse_n[h] * kLog2e;
        cf_u[h] = cf_n[h];
      }
      if (k + 1 < n_my) {
        const int un = first + (k + 1) * stride;
        request(un, un / nsb != tile);
      }
      CE6_LSTAMP(2);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int row = 32 * sb + 16 * h + c;
        f4 a, a1 = {0.f, 0.f, 0.f, 0.f};  // (two chains: a dependent MFMA waits out the previous one's passes)
        a = a1;
#define CE6_L1(p, q)                                        \
  a = mfma_bf(Er1[0][p], Hq[(2 * h + 0) * 3 + q], a);      \
  a1 = mfma_bf(Er1[1][p], Hq[(2 * h + 1) * 3 + q], a1);
        CE6_TERMS(CE6_L1)
#undef CE6_L1
        a += a1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (itx + 4 * g + r >= N) a[r] = ACATTN_NEG_INF;
        float m_w = ACATTN_NEG_INF, s_w = 0.f;
        if (DIR) {
          m_w = quad_max(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
          const float m2 = m_w > ACATTN_NEG_INF ? m_w * kLog2e : 0.f;
          f4 x = a * kLog2e - m2;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
          a = x;
          s_w = quad_sum((x[0] + x[1]) + (x[2] + x[3]));
        } else {
          f4 x = a * kLog2e - l2_u[h];
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
          a = x * cf_u[h];
        }
        float x8[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          x8[r] = a[r];
          x8[4 + r] = 0.f;
        }
        b8 dlB1[3];
        split8(x8, dlB1[0], dlB1[1], dlB1[2]);
        CE6_LSTAMP(3);
        if (WITH_TABLE_GRAD) {
#pragma unroll
          for (int p = 0; p < 3; ++p)
            *(b4*)(Xw + p * XPL1 + (16 * h + c) * XRS1 + 8 * g) = __builtin_shufflevector(dlB1[p], dlB1[p], 0, 1, 2, 3);
        }
        f4 dh[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) dh[cb] = f4{0.f, 0.f, 0.f, 0.f};
#define CE6_L2(p, q) \
  _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) dh[cb] = mfma_bf(Ec1[cb][p], dlB1[q], dh[cb]);
        CE6_TERMS(CE6_L2)
#undef CE6_L2
        if (row < B) {
#pragma unroll
          for (int cb = 0; cb < 4; ++cb) *(f4*)(d_out_slab + (vslab + row) * CH + 16 * cb + 4 * g) = dh[cb];
          if (DIR && g == 0) part[vslab + row] = float2{m_w, s_w};
        }
        CE6_LSTAMP(4);
      }
      if (WITH_TABLE_GRAD) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        b8 At[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const char* src = Xw + p * XPL1 + (8 * g + (c >> 2)) * XRS1 + 8 * (c & 3);
          const s4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(src));
          const s4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)(src + 4 * XRS1));
          At[p] = __builtin_bit_cast(b8, (s8v)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
#define CE6_L3(p, q) \
  _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) dE1[cb] = mfma_bf(At[p], Hq[12 + 4 * q + cb], dE1[cb]);
        CE6_TERMS(CE6_L3)
#undef CE6_L3
      }
      CE6_LSTAMP(5);
    }
    CE6_LSTAMP(-1);
    if (WITH_TABLE_GRAD && tile_mine < n_left) {  // the four waves' shares of the tile's d_table rows meet in the parked-tile areas
      const int itx = gridDim.x * NW * C::ITEMS + 16 * tile_mine;
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pw[(4 * g + r) * ES + 16 * cb + c] = dE1[cb][r];
      __syncthreads();
      for (int idx = threadIdx.x; idx < 16 * CH; idx += 64 * NW) {
        const int i = idx / CH, hcol = idx - i * CH;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += park[w * C::PARK_FLOATS + i * ES + hcol];
        if (itx + i < N) {
          if (pieces > 1) atomicAdd(d_table + (size_t)(itx + i) * CH + hcol, v);
          else d_table[(size_t)(itx + i) * CH + hcol] = v;
        }
      }
    }
  }
  CE6_STAMP(7);
#ifdef ACATTN_CE_STAMPS
  if (lane == 0 && blockIdx.x * NW + wave < 4096)
    for (int k = 0; k < 8; ++k) g_ce6_stamps[(blockIdx.x * NW + wave) * 8 + k] = cyc_[k];
#endif
}

// ---------------------------------------------------------------------------------------------------------
// forward: per (row, wave) partial (max, sum exp) of the wave's items (ce_fwd_reduce_kernel of acattn_ce.hip folds them)
// ---------------------------------------------------------------------------------------------------------
// P1 alone.  EIGHT waves of three tiles per workgroup (the same 384 items as the backward's four waves of six): the table
// operands are 72 registers, two waves share a SIMD and one's soft-max arithmetic runs under the other's MFMAs.  The
// rows' images (the 12 by-row slots) come by LDS-DMA into a double buffer, one barrier per super-block.
constexpr int NWF = 8;
constexpr int FT = 3;
constexpr int HR_BYTES = 12 * 64 * 16;

__global__ void __launch_bounds__(64 * NWF) ce6_fwd_kernel(const acattn_ce_problem P, float2* __restrict__ part,
                                                           const b8* __restrict__ Hb, const int n_left) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int c = lane & 15, g = lane >> 4;
  const int wid = blockIdx.x * NWF + wave;
  const int item0 = wid * 16 * FT;
  const int B = P.B, N = P.N;
  const bool ragged = item0 + 16 * FT > N;  // (uniform per wave)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][12][64] x 16 bytes, then the fold area

#ifdef ACATTN_CE_STAMPS
  unsigned long long cyc_[8] = {}, last_ = 0;
#endif
  CE6_STAMP(-1);
  b8 Er[FT][2][3];
#pragma unroll
  for (int t = 0; t < FT; ++t) {
    const int item = item0 + 16 * t + c;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float* src = P.table + (size_t)min(item, N - 1) * CH + 32 * s + 8 * g;
      const f4 v0 = *(const f4*)src, v1 = *(const f4*)(src + 4);
      float x[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[j] = item < N ? v0[j] : 0.f;
        x[4 + j] = item < N ? v1[j] : 0.f;
      }
      split8(x, Er[t][s][0], Er[t][s][1], Er[t][s][2]);
    }
  }
  const int nsb = (B + 31) >> 5;
  auto dma_rows = [&](int sb) {  // waves 0..5 two slots each
    if (wave < 6) {
      const char* src = (const char*)Hb + (size_t)sb * HB_BYTES + (2 * wave) * 1024 + lane * 16;
      char* dst = smem + (sb & 1) * HR_BYTES + (2 * wave) * 1024;
      dma16(src, dst);
      dma16(src + 1024, dst + 1024);
    }
  };
  dma_rows(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // Waves w and w + 4 share a SIMD and leave every barrier in step: both would want the matrix pipe, then both the vector
  // pipe.  With different priorities one runs ahead by a phase and each one's soft-max arithmetic falls under the other's
  // MFMAs.
  static_assert(NWF == 8, "two waves per SIMD");
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  // The eight waves' (max, sum-exp) pairs of a super-block's 32 rows meet in LDS ([2][NWF][32] pairs, by super-block
  // parity) and are folded by the first 32 threads at the top of the NEXT iteration, behind the barrier that ends this
  // one: one partial per (workgroup, row) leaves the kernel, and the wait in front of the barrier (for the DMA) does not
  // sit behind stores.
  float2* fold = (float2*)(smem + 2 * HR_BYTES);
  auto store_res = [&](int sb) {
    if (threadIdx.x < 32) {
      const float2* f = fold + (sb & 1) * NWF * 32 + threadIdx.x;
      float2 v[NWF];
      float m = ACATTN_NEG_INF;
#pragma unroll
      for (int w = 0; w < NWF; ++w) {
        v[w] = f[w * 32];
        m = fmaxf(m, v[w].x);
      }
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < NWF; ++w)
        sum += v[w].x > ACATTN_NEG_INF ? v[w].y * __builtin_amdgcn_exp2f((v[w].x - m) * kLog2e) : 0.f;
      const int row = 32 * sb + threadIdx.x;
      if (row < B) part[(size_t)blockIdx.x * B + row] = float2{m, sum};
    }
  };
  for (int sb = 0; sb < nsb; ++sb) {
    const b8* Hs = (const b8*)(smem + (sb & 1) * HR_BYTES);
    if (sb + 1 < nsb) dma_rows(sb + 1);  // its buffer was last read before the previous super-block's barrier
    if (sb > 0) store_res(sb - 1);
    CE6_STAMP(sb == 0 ? 6 : 3);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      b8 Hr[2][3];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 3; ++q) Hr[s][q] = Hs[((2 * h + s) * 3 + q) * 64 + lane];
      f4 acc[FT];
#pragma unroll
      for (int t = 0; t < FT; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
#define CE6_F1(p, q)                             \
  _Pragma("unroll") for (int s = 0; s < 2; ++s) \
      _Pragma("unroll") for (int t = 0; t < FT; ++t) acc[t] = mfma_bf(Er[t][s][p], Hr[s][q], acc[t]);
      CE6_TERMS(CE6_F1)
#undef CE6_F1
      CE6_STAMP(0);
      if (ragged) {
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (item0 + 16 * t + 4 * g + r >= N) acc[t][r] = ACATTN_NEG_INF;
      }
      float m = ACATTN_NEG_INF;
#pragma unroll
      for (int t = 0; t < FT; ++t) m = fmaxf(fmaxf(fmaxf(fmaxf(m, acc[t][0]), acc[t][1]), acc[t][2]), acc[t][3]);
      m = quad_max(m);
      float sum;
      {
        const float m2 = m > ACATTN_NEG_INF ? m * kLog2e : 0.f;  // (no item: exp2(-inf) = 0, no branch)
        f4 sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < FT; ++t) {
          f4 x = acc[t] * kLog2e - m2;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
          sv += x;
        }
        sum = (sv[0] + sv[1]) + (sv[2] + sv[3]);
      }
      sum = quad_sum(sum);
      if (g == 0) fold[((sb & 1) * NWF + wave) * 32 + 16 * h + c] = float2{m, sum};
      CE6_STAMP(1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CE6_STAMP(2);
  }
  store_res(nsb - 1);
  if (n_left > 0) {
    // leftover units (tile, super-block), n_left * nsb of them, dealt out evenly: wave w takes units [w U, (w + 1) U);
    // operand images straight from global memory.  Partial number gridDim.x + tile.
    const int n_units = n_left * nsb, U = (n_units + gridDim.x * NWF - 1) / (gridDim.x * NWF);
    const int first = wid * U, n_my = min(max(n_units - first, 0), U);
    int cur_tile = -1;
    b8 Er1[2][3];
    for (int k = 0; k < n_my; ++k) {
      const int u = first + k, tile = u / nsb, sb = u - tile * nsb;
      const int itx = gridDim.x * NWF * 16 * FT + 16 * tile;
      b8 Hq[12];
      const b8* src = Hb + (size_t)sb * HSLOTS * 64 + lane;
#pragma unroll
      for (int i = 0; i < 12; ++i) Hq[i] = src[i * 64];
      if (tile != cur_tile) {  // (uniform per wave)
        cur_tile = tile;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const float* tsrc = P.table + (size_t)min(itx + c, N - 1) * CH + 32 * s + 8 * g;
          const f4 v0 = *(const f4*)tsrc, v1 = *(const f4*)(tsrc + 4);
          float x[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            x[j] = itx + c < N ? v0[j] : 0.f;
            x[4 + j] = itx + c < N ? v1[j] : 0.f;
          }
          split8(x, Er1[s][0], Er1[s][1], Er1[s][2]);
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        f4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
#define CE6_FL(p, q)                                          \
  a0 = mfma_bf(Er1[0][p], Hq[(2 * h + 0) * 3 + q], a0);      \
  a1 = mfma_bf(Er1[1][p], Hq[(2 * h + 1) * 3 + q], a1);
        CE6_TERMS(CE6_FL)
#undef CE6_FL
        f4 a = a0 + a1;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (itx + 4 * g + r >= N) a[r] = ACATTN_NEG_INF;
        const float m = quad_max(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
        float sum = 0.f;
        if (m > ACATTN_NEG_INF) {
          f4 x = a * kLog2e - m * kLog2e;
#pragma unroll
          for (int r = 0; r < 4; ++r) x[r] = __builtin_amdgcn_exp2f(x[r]);
          sum = (x[0] + x[1]) + (x[2] + x[3]);
        }
        sum = quad_sum(sum);
        const int row = 32 * sb + 16 * h + c;
        if (g == 0 && row < B) part[(size_t)(gridDim.x + tile) * B + row] = float2{m, sum};
      }
    }
  }
  CE6_STAMP(7);
#ifdef ACATTN_CE_STAMPS
  if (lane == 0 && wid < 4096)
    for (int k = 0; k < 8; ++k) g_ce6_stamps[wid * 8 + k] = cyc_[k];
#endif
}

// d_out[i] = sum over workgroups of slab[wg][i] (acattn_ce.hip's ce_bwd_reduce_kernel: 32 outputs per workgroup, 8 threads
// per output) - coef_row E_target(row)[ch], and d_table[target(row)][ch] -= coef_row out[row][ch]: the one-hot part of
// dl = coef (softmax - onehot), which the sweep leaves out.  d_table's rows were written by the sweep in front of this
// launch; several rows may share a target, hence atomics (B x 64 of them).
__global__ void __launch_bounds__(256) ce6_onehot_reduce_kernel(const acattn_ce_problem P, const float* __restrict__ coef,
                                                                 const float* __restrict__ slab, const int n_slabs,
                                                                 float* __restrict__ d_out, float* __restrict__ d_table) {
  const int sl = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int64_t n_out = (int64_t)P.B * CH;
  const int64_t i = (int64_t)blockIdx.x * 32 + sl;
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  if (i < n_out) {
    const float* ps = slab + i;
    int p = q;
    for (; p + 56 < n_slabs; p += 64) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += ps[(size_t)(p + 8 * u) * n_out];
    }
    for (; p < n_slabs; p += 8) a[0] += ps[(size_t)p * n_out];
  }
  __shared__ float red[8][32];
  red[q][sl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (q == 0 && i < n_out) {
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) v += red[u][sl];
    const int row = (int)(i / CH), ch = (int)(i - (int64_t)row * CH);
    const long long tgt = P.target[row];
    if (tgt >= 0 && tgt < P.N) {  // (an invalid target: the row's loss is NaN already, see ce_fwd_reduce_kernel)
      const float cf = coef[P.coef_is_scalar ? 0 : row] * (P.coef_scale != 0.f ? P.coef_scale : 1.0f);
      v -= cf * P.table[(size_t)tgt * CH + ch];
      if (d_table) atomicAdd(d_table + (size_t)tgt * CH + ch, -cf * P.out[i]);
    }
    d_out[i] = v;
  }
}

template <class K>
void allow_lds(K k, size_t lds) {
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

}  // namespace

// Bytes of the batch rows' operand images (behind the regular workspace of acattn_ce.hip).
int64_t acattn_ce6_rows_bytes(const acattn_ce_problem& p) { return (int64_t)((p.B + 31) / 32) * HB_BYTES; }

// n_wg workgroups of 4 waves x 6 tiles (+ n_left leftover tiles); slab / part as in acattn_ce.hip's launchers.
int acattn_launch_ce6_sweep(const acattn_ce_problem& p, const float* lse, const float* coef, float* slab, float* d_table,
                            float2* part, void* rows_ws, int n_wg, int n_left, bool dir, hipStream_t stream) {
  using C = Ce6<6>;
  b8* Hb = (b8*)rows_ws;
  float* zero = nullptr;
  int64_t n_zero = 0;
  if (d_table && !dir && n_left > 0 && n_wg / n_left > 1) {  // (the kernel's `pieces` > 1)
    const int64_t first_item = (int64_t)n_wg * NW * C::ITEMS;
    zero = d_table + first_item * CH;
    n_zero = (p.N - first_item) * CH;
  }
  hipLaunchKernelGGL(ce_split_rows_kernel, dim3((p.B + 31) / 32), dim3(256), 0, stream, p.out, p.B, Hb, zero, n_zero);
  const size_t lds = C::LDS_BYTES;
  if (dir) {
    auto k = ce6_bwd_kernel<6, false, true>;
    allow_lds(k, lds);
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * NW), lds, stream, p, lse, coef, slab, d_table, part, (const b8*)Hb, n_left);
  } else if (d_table) {
    auto k = ce6_bwd_kernel<6, true, false>;
    allow_lds(k, lds);
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * NW), lds, stream, p, lse, coef, slab, d_table, part, (const b8*)Hb, n_left);
  } else {
    auto k = ce6_bwd_kernel<6, false, false>;
    allow_lds(k, lds);
    hipLaunchKernelGGL(k, dim3(n_wg), dim3(64 * NW), lds, stream, p, lse, coef, slab, d_table, part, (const b8*)Hb, n_left);
  }
  return (int)hipGetLastError();
}

#ifdef ACATTN_CE_STAMPS
extern "C" int acattn_debug_ce6_stamps(unsigned long long* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ce6_stamps), (size_t)n_words * 8);
}
#endif

int acattn_launch_ce6_onehot_reduce(const acattn_ce_problem& p, const float* coef, const float* slab, int n_slabs, float* d_out,
                                    float* d_table, hipStream_t stream) {
  const int64_t n_out = (int64_t)p.B * CH;
  hipLaunchKernelGGL(ce6_onehot_reduce_kernel, dim3((unsigned)((n_out + 31) / 32)), dim3(256), 0, stream, p, coef, slab, n_slabs,
                     d_out, d_table);
  return (int)hipGetLastError();
}

// Forward partials: (n_wg + n_left) x B (max, sum-exp) pairs in `part`.
int acattn_launch_ce6_fwd_sweep(const acattn_ce_problem& p, float2* part, void* rows_ws, int n_wg, int n_left, hipStream_t stream) {
  b8* Hb = (b8*)rows_ws;
  hipLaunchKernelGGL(ce_split_rows_kernel, dim3((p.B + 31) / 32), dim3(256), 0, stream, p.out, p.B, Hb, (float*)nullptr, (int64_t)0);
  hipLaunchKernelGGL(ce6_fwd_kernel, dim3(n_wg), dim3(64 * NWF), 2 * HR_BYTES + 2 * NWF * 32 * sizeof(float2), stream, p, part, (const b8*)Hb, n_left);
  return (int)hipGetLastError();
}
