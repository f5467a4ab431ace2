// Weight and bias gradients of y = x W^T + b for tall-skinny activations, gfx950.
//
//     dW[N, K] = sum_m dy[m, n] x[m, k],   db[n] = sum_m dy[m, n],   M = B*L rows (25,600 at the benchmark shape)
//
// These are the parameter gradients of every projection around the calibrated-attention core (query/key/value,
// the two attack transforms, dense, the gate, the feed-forward pair: recbole/model/layers.py:687-690, 660-661, 681,
// 792-794, 863).  The reduction runs over M while the output is at most 256 x 256: a GEMM shape BLAS libraries
// answer badly (hipBLASLt stream-K: 110-125 us; a batched split-K out of library calls + two reduction launches
// for the slabs + two for the bias: ~25 us), although the work is one pass over 2 x 6.5 MB.
//
// Stage 1 (wgrad_partial_kernel): workgroup (p, block) owns a 64(n) x 64(k) block of dW and every P-th slab of
// rows.  A wave reads 4 rows x 64 columns of dy and of x with ONE coalesced 16-byte load each (lane 16g+c: row
// g, columns 4c..4c+3) and feeds them straight into fp32 MFMAs: in v_mfma_f32_16x16x4_f32 lane 16g+c supplies
// A[c][g] and B[g][c], so register e' of the dy load is the A operand of the n-columns {4i+e'} and register e of
// the x load is the B operand of the k-columns {4j+e}: 16 MFMAs per 4 rows cover the whole 64 x 64 block with
// no shuffles and no LDS; the block is simply held in a column-permuted order until the final store.
// Waves of a workgroup are folded through LDS, the workgroup writes one partial.
// Stage 2 (wgrad_reduce_kernel): sums the P partials in a fixed order (deterministic, no atomics), undoes the
// permutation, writes dW and db.
#include <stdlib.h>

#include <algorithm>

#include "acattn_common.h"
#include "acattn_sumrows.h"

namespace {

constexpr int kRegs = 64;  // accumulator registers per lane: 16 MFMA tiles x 4

// Up to ACATTN_WGRAD_MAX_GROUP problems of one launch: same M, own K, N and operands.  blockIdx.z selects the
// item; grid.y covers the largest item's 64 x 64 blocks (workgroups beyond an item's own block count leave at once);
// an item's partials live in its own slice of the workspace (offsets in floats).
struct WgradGroup {
  const float* x[ACATTN_WGRAD_MAX_GROUP];
  const float* dy[ACATTN_WGRAD_MAX_GROUP];
  float* dw[ACATTN_WGRAD_MAX_GROUP];
  float* db[ACATTN_WGRAD_MAX_GROUP];
  int K[ACATTN_WGRAD_MAX_GROUP];
  int N[ACATTN_WGRAD_MAX_GROUP];
  long long w_off[ACATTN_WGRAD_MAX_GROUP];
  long long b_off[ACATTN_WGRAD_MAX_GROUP];
};

__device__ __forceinline__ f4 load_rows4(const float* base, int64_t row, int64_t M, int ld, int col, int ncols) {
  // 4 consecutive columns of `row`, zero outside the matrix; rows of a matrix with ld % 4 != 0 are only
  // dword-aligned, and its last columns must not be read past the row end
  f4 v = {0.f, 0.f, 0.f, 0.f};
  if (row < M) {
    const float* p = base + row * ld + col;
    if (col + 3 < ncols) {
      v = *(const f4u*)p;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < ncols) v[e] = p[e];
    }
  }
  return v;
}

// (256, 2): the 64 KB of LDS allow two workgroups per CU; the register budget must too
template <int UNROLL>
__global__ void __launch_bounds__(256, 2) wgrad_partial_kernel(const WgradGroup G, const int64_t M,
                                                             float* __restrict__ ws) {
  const int it = blockIdx.z;
  const int K = G.K[it], N = G.N[it];
  const int KB = (K + 63) >> 6, NB = (N + 63) >> 6;
  if ((int)blockIdx.y >= KB * NB) return;  // (uniform per workgroup, before any barrier)
  const float* __restrict__ x = G.x[it];
  const float* __restrict__ dy = G.dy[it];
  float* part_w = ws + G.w_off[it];
  float* part_b = G.db[it] ? ws + G.b_off[it] : nullptr;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int blk = blockIdx.y, nb = blk / KB, kb = blk - nb * KB;
  const int P = gridDim.x;
  const int ncol = nb * 64 + 4 * c, kcol = kb * 64 + 4 * c;

  f4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  f4 bsum = {0.f, 0.f, 0.f, 0.f};

  // row groups of 4; wave w of workgroup p takes groups (p * 4 + w) + i * 4P, UNROLL groups per batch
  const int64_t stride = (int64_t)P * 4 * 4, batch = stride * UNROLL;
  int64_t m0 = ((int64_t)blockIdx.x * 4 + wave) * 4;
  if ((K & 63) == 0 && (N & 63) == 0) {  // (uniform per workgroup)
    // Every block lies inside its matrix: plain 16-byte loads, a row past the end is read as row M - 1 and its dy
    // zeroed when consumed (a select right behind the load would wait for it).  Two batch buffers: the next batch's
    // loads are requested before the current batch's MFMAs (pinned: the scheduler would sink them behind the MFMAs
    // again), so a wave pays one memory latency per launch, not one per batch.
    auto load_batch = [&](int64_t r0, f4 (&xv)[UNROLL], f4 (&gv)[UNROLL]) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const int64_t row = min(r0 + u * stride + g, M - 1);
        xv[u] = *(const f4*)(x + row * K + kcol);
        gv[u] = *(const f4*)(dy + row * N + ncol);
      }
    };
    auto mma_batch = [&](int64_t r0, const f4 (&xv)[UNROLL], const f4 (&gv)[UNROLL]) {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        const bool row_in = r0 + u * stride + g < M;
        f4 gq;
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[e] = row_in ? gv[u][e] : 0.f;
        bsum += gq;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(gq[a], xv[u][b], acc[a][b]);
      }
    };
    f4 xa[UNROLL], ga[UNROLL], xb[UNROLL], gb[UNROLL];
    load_batch(m0, xa, ga);
    // two batches per trip, no exit in between (an exit there costs a second copy of the 64 accumulators)
#pragma nounroll
    for (; m0 < M; m0 += 2 * batch) {
      load_batch(m0 + batch, xb, gb);
      __builtin_amdgcn_sched_barrier(0);
      mma_batch(m0, xa, ga);
      load_batch(m0 + 2 * batch, xa, ga);
      __builtin_amdgcn_sched_barrier(0);
      if (m0 + batch < M) mma_batch(m0 + batch, xb, gb);  // (wave-uniform)
    }
  } else {
    // a block sticks out of its matrix (the gate: N = seq_length) or rows are only dword-aligned: checked loads,
    // one batch at a time
    for (; m0 < M; m0 += batch) {
      f4 xv[UNROLL], gv[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {  // all loads of the iteration in flight before the first MFMA
        const int64_t row = m0 + u * stride + g;
        xv[u] = load_rows4(x, row, M, K, kcol, K);
        gv[u] = load_rows4(dy, row, M, N, ncol, N);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        bsum += gv[u];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = mfma16(gv[u][a], xv[u][b], acc[a][b]);
      }
    }
  }

  // fold the 4 waves: [wave][reg][lane] in LDS, thread t sums the 4 copies of 16 (reg, lane) slots
  __shared__ float red[4 * kRegs * 64];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * kRegs + (a * 4 + b) * 4 + r) * 64 + lane] = acc[a][b][r];
  __syncthreads();
  float* pw = part_w + ((size_t)blk * P + blockIdx.x) * (kRegs * 64);
  for (int s = threadIdx.x; s < kRegs * 64; s += 256)
    pw[s] = (red[s] + red[kRegs * 64 + s]) + (red[2 * kRegs * 64 + s] + red[3 * kRegs * 64 + s]);
  if (part_b && kb == 0) {
    __syncthreads();
    *(f4*)(red + 4 * threadIdx.x) = bsum;  // [wave][g][c][e]
    __syncthreads();
    if (threadIdx.x < 64) {  // column nb*64 + t = 4c + e  ->  c = t >> 2, e = t & 3
      float s = 0.f;
#pragma unroll
      for (int wg = 0; wg < 16; ++wg) s += red[4 * (wg * 16 + (threadIdx.x >> 2)) + (threadIdx.x & 3)];
      part_b[((size_t)nb * P + blockIdx.x) * 64 + threadIdx.x] = s;
    }
  }
}

// dW[n, k] for n = nb*64 + 16g + 4r + a, k = kb*64 + 4c + b lives at slot ((a*4 + b)*4 + r) * 64 + 16g + c of
// every partial (D register r of lane 16g+c is D[4g + r][c]; tile (a, b) holds n-columns {4i + a}, k-columns {4j + b}).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const WgradGroup G, const float* __restrict__ ws,
                                                            const int P) {
  const int it = blockIdx.z;
  const int K = G.K[it], N = G.N[it];
  const int KB = (K + 63) >> 6, NB = (N + 63) >> 6;
  if ((int)blockIdx.y >= KB * NB) return;
  float* __restrict__ dw = G.dw[it];
  float* __restrict__ db = G.db[it];
  const float* part_w = ws + G.w_off[it];
  const float* part_b = ws + G.b_off[it];
  const int blk = blockIdx.y, nb = blk / KB, kb = blk - nb * KB;
  // 32 slots per workgroup, 8 threads per slot; a thread sums every 8th partial with 8 loads in flight at a time
  // (the kernel is a latency chain otherwise: 4 MB spread over few workgroups), folded through LDS
  const int sl = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int s = blockIdx.x * 32 + sl;  // slot 0..4095
  const float* pw = part_w + (size_t)blk * P * (kRegs * 64) + s;
  constexpr size_t PS = kRegs * 64;
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  int p = q;
  for (; p + 56 < P; p += 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += pw[(size_t)(p + 8 * u) * PS];
  }
  for (; p < P; p += 8) a[0] += pw[(size_t)p * PS];
  __shared__ float red[8][32];
  red[q][sl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (q == 0) {
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) v += red[u][sl];
    const int reg = s >> 6, lane = s & 63;
    const int tile = reg >> 2, r = reg & 3, aa = tile >> 2, bb = tile & 3, g = lane >> 4, c = lane & 15;
    const int n = nb * 64 + 4 * (4 * g + r) + aa, k = kb * 64 + 4 * c + bb;
    if (n < N && k < K) dw[(size_t)n * K + k] = v;
  }
  if (db && kb == 0 && blockIdx.x < 2) {  // 2 workgroups x 32 columns
    __syncthreads();
    const int col = blockIdx.x * 32 + sl, nn = nb * 64 + col;
    const float* pb = part_b + (size_t)nb * P * 64 + col;
    float sb = 0.f;
    for (int t = q; t < P; t += 8) sb += pb[(size_t)t * 64];
    red[q][sl] = sb;
    __syncthreads();
    if (q == 0 && nn < N) {
      float v = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) v += red[u][sl];
      db[nn] = v;
    }
  }
}

// Stage 2 for up to ACATTN_WGRAD_MAX_REDUCE items of DIFFERENT launches of stage 1 (each with its own partial count and
// workspace): a trainer defers the reductions of a whole backward walk to one launch (acattn_linear_wgrad_reduce_many).
struct WgradMany {
  const float* part_w[ACATTN_WGRAD_MAX_REDUCE];
  const float* part_b[ACATTN_WGRAD_MAX_REDUCE];
  float* dw[ACATTN_WGRAD_MAX_REDUCE];
  float* db[ACATTN_WGRAD_MAX_REDUCE];
  int K[ACATTN_WGRAD_MAX_REDUCE];
  int N[ACATTN_WGRAD_MAX_REDUCE];
  int P[ACATTN_WGRAD_MAX_REDUCE];
  // ... and plain row sums out[c] = sum_r x[r, c] of the same walk (LayerNorm / calibrator parameter partials), as further
  // z-slices of the launch: blockIdx.z >= n_w
  SumRowsJob sr[ACATTN_SUMROWS_MAX_DEFER];
  int n_w;
};

__global__ void __launch_bounds__(256) wgrad_reduce_many_kernel(const WgradMany G) {
  if ((int)blockIdx.z >= G.n_w) {  // (uniform per workgroup)
    const SumRowsJob& j = G.sr[blockIdx.z - G.n_w];
    const int w = blockIdx.y * gridDim.x + blockIdx.x;
    if (w < j.n_wg) sum_rows_block<false>(j.x, j.out, j.R, j.C, j.R, j.CT, w % j.col_groups, 0, w / j.col_groups);
    return;
  }
  const int it = blockIdx.z;
  const int K = G.K[it], N = G.N[it];
  const int KB = (K + 63) >> 6, NB = (N + 63) >> 6;
  if ((int)blockIdx.y >= KB * NB) return;
  float* __restrict__ dw = G.dw[it];
  float* __restrict__ db = G.db[it];
  const float* part_w = G.part_w[it];
  const float* part_b = G.part_b[it];
  const int P = G.P[it];
  const int blk = blockIdx.y, nb = blk / KB, kb = blk - nb * KB;
  // 32 slots per workgroup, 8 threads per slot; a thread sums every 8th partial with 8 loads in flight at a time
  // (the kernel is a latency chain otherwise: 4 MB spread over few workgroups), folded through LDS
  const int sl = threadIdx.x & 31, q = threadIdx.x >> 5;
  const int s = blockIdx.x * 32 + sl;  // slot 0..4095
  const float* pw = part_w + (size_t)blk * P * (kRegs * 64) + s;
  constexpr size_t PS = kRegs * 64;
  float a[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) a[u] = 0.f;
  int p = q;
  for (; p + 56 < P; p += 64) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] += pw[(size_t)(p + 8 * u) * PS];
  }
  for (; p < P; p += 8) a[0] += pw[(size_t)p * PS];
  __shared__ float red[8][32];
  red[q][sl] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  __syncthreads();
  if (q == 0) {
    float v = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) v += red[u][sl];
    const int reg = s >> 6, lane = s & 63;
    const int tile = reg >> 2, r = reg & 3, aa = tile >> 2, bb = tile & 3, g = lane >> 4, c = lane & 15;
    const int n = nb * 64 + 4 * (4 * g + r) + aa, k = kb * 64 + 4 * c + bb;
    if (n < N && k < K) dw[(size_t)n * K + k] = v;
  }
  if (db && kb == 0 && blockIdx.x < 2) {  // 2 workgroups x 32 columns
    __syncthreads();
    const int col = blockIdx.x * 32 + sl, nn = nb * 64 + col;
    const float* pb = part_b + (size_t)nb * P * 64 + col;
    float sb = 0.f;
    for (int t = q; t < P; t += 8) sb += pb[(size_t)t * 64];
    red[q][sl] = sb;
    __syncthreads();
    if (q == 0 && nn < N) {
      float v = 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u) v += red[u][sl];
      db[nn] = v;
    }
  }
}

int pick_partials(int64_t M, int blocks) {
  // Measured on MI355X at M = 25,600 (tools/gpu_wgrad.sh): stage 1 takes ~10 us whether 128 or 256 workgroups
  // share a 64 x 64 block, stage 2 grows with the number of partials it folds -> few partials, but never fewer
  // than 64 workgroups per block and never less than 64 rows per workgroup.
  static const int forced = getenv("ACATTN_WGRAD_PARTIALS") ? atoi(getenv("ACATTN_WGRAD_PARTIALS")) : 0;  // measurements
  int p = forced > 0 ? forced : std::max(64, 128 / blocks);
  return (int)std::min<int64_t>(p, std::max<int64_t>(1, M / 64));
}

}  // namespace

namespace {
int64_t item_ws_floats(int K, int N, int P) {
  const int64_t KB = (K + 63) / 64, NB = (N + 63) / 64;
  return KB * NB * P * kRegs * 64 + NB * P * 64;
}
int group_partials(int64_t M, const int* K, const int* N, int n_items) {
  int blocks = 1;
  for (int i = 0; i < n_items; ++i) blocks = std::max(blocks, ((K[i] + 63) / 64) * ((N[i] + 63) / 64));
  return pick_partials(M, blocks);
}
}  // namespace

// upper bound for any group this problem may be launched in: P never exceeds pick_partials(M, 1)
int64_t acattn_linear_wgrad_ws_bytes(int64_t M, int K, int N) {
  return item_ws_floats(K, N, pick_partials(M, 1)) * (int64_t)sizeof(float);
}

// n_items problems sharing M (the workspace holds sum_i acattn_linear_wgrad_ws_bytes(M, K[i], N[i]) bytes)
int acattn_launch_linear_wgrad(const float* const* x, const float* const* dy, const int* K, const int* N,
                               float* const* dw, float* const* db, int n_items, int64_t M, void* ws, hipStream_t stream,
                               int* P_out, long long* w_off_out, long long* b_off_out) {
  WgradGroup G{};
  const int P = group_partials(M, K, N, n_items);
  int blocks = 1;
  long long off = 0;
  for (int i = 0; i < n_items; ++i) {
    G.x[i] = x[i];
    G.dy[i] = dy[i];
    G.dw[i] = dw[i];
    G.db[i] = db[i];
    G.K[i] = K[i];
    G.N[i] = N[i];
    const long long KB = (K[i] + 63) / 64, NB = (N[i] + 63) / 64;
    G.w_off[i] = off;
    G.b_off[i] = off + KB * NB * P * kRegs * 64;
    off += item_ws_floats(K[i], N[i], P);
    blocks = std::max(blocks, (int)(KB * NB));
  }
  // a wave's loads are all issued before its first MFMA when they fit (one HBM latency instead of several)
  const int64_t groups_per_wave = ((M + 3) / 4 + (int64_t)P * 4 - 1) / ((int64_t)P * 4);
  const dim3 grid(P, blocks, n_items);
  if (groups_per_wave > 4)
    hipLaunchKernelGGL((wgrad_partial_kernel<5>), grid, dim3(256), 0, stream, G, M, (float*)ws);
  else if (groups_per_wave >= 4)
    hipLaunchKernelGGL((wgrad_partial_kernel<4>), grid, dim3(256), 0, stream, G, M, (float*)ws);
  else
    hipLaunchKernelGGL((wgrad_partial_kernel<1>), grid, dim3(256), 0, stream, G, M, (float*)ws);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  if (P_out) {  // stage 1 alone: the caller reduces later (acattn_launch_linear_wgrad_reduce_many)
    *P_out = P;
    for (int i = 0; i < n_items; ++i) {
      w_off_out[i] = G.w_off[i];
      b_off_out[i] = G.b_off[i];
    }
    return 0;
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(kRegs * 2, blocks, n_items), dim3(256), 0, stream, G, (const float*)ws, P);
  return (int)hipGetLastError();
}

int acattn_launch_linear_wgrad_reduce_many(const float* const* part_w, const float* const* part_b, const int* K, const int* N,
                                           const int* P, float* const* dw, float* const* db, int n_items,
                                           const float* const* sr_x, float* const* sr_out, const int* sr_R, const int* sr_C,
                                           int n_sr, hipStream_t stream) {
  WgradMany G{};
  G.n_w = n_items;
  int blocks = 1;
  for (int i = 0; i < n_sr; ++i) {
    G.sr[i] = make_job(sr_x[i], sr_out[i], 1, sr_R[i], sr_C[i]);
    blocks = std::max(blocks, (G.sr[i].n_wg + kRegs * 2 - 1) / (kRegs * 2));
  }
  for (int i = 0; i < n_items; ++i) {
    G.part_w[i] = part_w[i];
    G.part_b[i] = part_b[i];
    G.dw[i] = dw[i];
    G.db[i] = db[i];
    G.K[i] = K[i];
    G.N[i] = N[i];
    G.P[i] = P[i];
    blocks = std::max(blocks, ((K[i] + 63) / 64) * ((N[i] + 63) / 64));
  }
  hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3(kRegs * 2, blocks, n_items + n_sr), dim3(256), 0, stream, G);
  return (int)hipGetLastError();
}
