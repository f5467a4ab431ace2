// torch.optim.Adam's update for every parameter of the model in ONE launch (recbole/trainer/trainer.py:590-615 builds
// optim.Adam; the step is part of the measured training step).
//
// The step is a stream over 7 x 4 bytes per parameter element -- 180 MB with the 100k x 64 item table, whose CE gradient is
// dense.  torch's fused multi-tensor implementation takes two launches for the model's ~60 tensors plus one that
// increments the step counters: 55 + 14 + 5 us.  Here: one launch, tensor descriptors in the kernel arguments (a scalar
// scan finds a workgroup's tensor), 16-byte accesses, the step counters incremented by the last workgroup to finish.
//
// The arithmetic follows ATen/native/cuda/fused_adam_utils.cuh (adam_math, ADAM_MODE::ORIGINAL, no amsgrad, no maximize)
// operation by operation, including which products are formed in double (beta1, beta2, lr, eps, weight_decay are doubles
// there and the moments are floats), so that results agree with torch's to the last bit or two
// (tests/test_hip_adam.py).
#include <math.h>

#include <algorithm>

#include "acattn_common.h"

namespace {

constexpr int kChunk = 4096;  // elements per workgroup (2048: 75 us; four pipelined trips of 4096 per workgroup: 64 us;
                              // this: 62 us = 2.9 TB/s over the 180 MB of the benchmark model, torch's three launches: 74 us)

struct AdamArgs {
  acattn_adam_group g;
  int block_start[ACATTN_ADAM_MAX_TENSORS + 1];
  double lr, beta1, beta2, eps, weight_decay;
  int* done;
};

__device__ __forceinline__ void adam_math(float& param, float grad, float& exp_avg, float& exp_avg_sq, const AdamArgs& A,
                                          float bias_correction1, float bias_correction2_sqrt) {
  if (A.weight_decay != 0) grad += param * A.weight_decay;
  exp_avg = A.beta1 * exp_avg + (1 - A.beta1) * grad;
  exp_avg_sq = A.beta2 * exp_avg_sq + (1 - A.beta2) * grad * grad;
  const float step_size = A.lr / bias_correction1;
  const float denom = (sqrtf(exp_avg_sq) / bias_correction2_sqrt) + A.eps;
  param -= step_size * exp_avg / denom;
}

__global__ void __launch_bounds__(256) adam_step_kernel(const AdamArgs A) {
  int t = 0;
  while (t + 1 < A.g.n_tensors && (int)blockIdx.x >= A.block_start[t + 1]) ++t;  // (scalar: uniform per workgroup)
  const int64_t n = A.g.numel[t];
  const int64_t base = (int64_t)((int)blockIdx.x - A.block_start[t]) * kChunk;
  float* __restrict__ p = A.g.param[t];
  const float* __restrict__ gr = A.g.grad[t];
  float* __restrict__ m = A.g.exp_avg[t];
  float* __restrict__ v = A.g.exp_avg_sq[t];
  // torch increments the count first (_foreach_add(state_steps, 1)) and corrects with the incremented one
  // (every lane computes the two double-precision pow() itself, under the loads below: one lane per workgroup + a barrier
  // put ~4 us of serial latency in front of every workgroup)
  const float step = *A.g.step[t] + 1.0f;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(gr) | reinterpret_cast<uintptr_t>(m) |
                         reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  auto corrections = [&](float& bc1, float& bc2s) {
    bc1 = (float)(1 - pow(A.beta1, (double)step));
    bc2s = (float)sqrt(1 - pow(A.beta2, (double)step));
  };
  float bc1, bc2s;
  if (aligned && base + kChunk <= n) {
    constexpr int NV = kChunk / 1024;
    f4 pv[NV], gv[NV], mv[NV], vv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {  // every request of the workgroup in flight before the first use
      const int64_t i = base + (int64_t)(threadIdx.x + 256 * k) * 4;
      pv[k] = *(const f4*)(p + i);
      gv[k] = *(const f4*)(gr + i);
      mv[k] = *(const f4*)(m + i);
      vv[k] = *(const f4*)(v + i);
    }
    corrections(bc1, bc2s);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pv[k][e], me = mv[k][e], ve = vv[k][e];
        adam_math(pe, gv[k][e], me, ve, A, bc1, bc2s);
        pv[k][e] = pe, mv[k][e] = me, vv[k][e] = ve;
      }
      const int64_t i = base + (int64_t)(threadIdx.x + 256 * k) * 4;
      *(f4*)(p + i) = pv[k];
      *(f4*)(m + i) = mv[k];
      *(f4*)(v + i) = vv[k];
    }
  } else {
    corrections(bc1, bc2s);
    for (int64_t i = base + threadIdx.x; i < n && i < base + kChunk; i += 256) {
      float pe = p[i], me = m[i], ve = v[i];
      adam_math(pe, gr[i], me, ve, A, bc1, bc2s);
      p[i] = pe;
      m[i] = me;
      v[i] = ve;
    }
  }
  // The last workgroup to finish advances the counters: every workgroup has read its counter by then (its lanes read it
  // at their start, the barrier below waits for them, then thread 0 counts the workgroup in), and the next launch sees counters and `done` through the
  // kernel boundary.  No fence: a device-scope fence writes the XCD's L2 back on this part, once per workgroup (it made
  // the launch 2x slower than the three launches it replaces).
  __syncthreads();
  if (threadIdx.x == 0) {
    if (atomicAdd(A.done, 1) == (int)gridDim.x - 1) {
      for (int k = 0; k < A.g.n_tensors; ++k) *A.g.step[k] = *A.g.step[k] + 1.0f;
      *A.done = 0;
    }
  }
}

}  // namespace

int acattn_launch_adam_step(const acattn_adam_group& g, double lr, double beta1, double beta2, double eps,
                            double weight_decay, int* done, hipStream_t stream) {
  AdamArgs A;
  // largest tensors first: a workgroup finds its tensor by a scalar scan over block_start, and nearly all workgroups
  // belong to the item table (with the table last in the list the scan cost 70 us per launch)
  int order[ACATTN_ADAM_MAX_TENSORS];
  for (int t = 0; t < g.n_tensors; ++t) order[t] = t;
  std::stable_sort(order, order + g.n_tensors, [&](int a, int b) { return g.numel[a] > g.numel[b]; });
  A.g.n_tensors = g.n_tensors;
  int blocks = 0;
  for (int t = 0; t < g.n_tensors; ++t) {
    const int s = order[t];
    A.g.param[t] = g.param[s], A.g.grad[t] = g.grad[s], A.g.exp_avg[t] = g.exp_avg[s], A.g.exp_avg_sq[t] = g.exp_avg_sq[s];
    A.g.step[t] = g.step[s], A.g.numel[t] = g.numel[s];
    A.block_start[t] = blocks;
    blocks += (int)((g.numel[s] + kChunk - 1) / kChunk);
  }
  A.block_start[g.n_tensors] = blocks;
  A.lr = lr, A.beta1 = beta1, A.beta2 = beta2, A.eps = eps, A.weight_decay = weight_decay;
  A.done = done;
  if (blocks == 0) return 0;
  hipLaunchKernelGGL(adam_step_kernel, dim3(blocks), dim3(256), 0, stream, A);
  return (int)hipGetLastError();
}
