// Dispatch of acattn_calibrated_attention_fwd: streaming kernel, LDS-staged kernels, then the general kernel
// (acattn_fwd_general.inc, one translation unit per head size: acattn_fwd_dh16.hip ... acattn_fwd_dh128.hip); and the
// kernel that materialises the counter-mode randomness.
#include <stdlib.h>

#include "acattn_common.h"

int acattn_launch_fwd_general_dh16(const acattn_problem& p, const acattn_fwd_out& o, bool full, hipStream_t stream);
int acattn_launch_fwd_general_dh32(const acattn_problem& p, const acattn_fwd_out& o, bool full, hipStream_t stream);
int acattn_launch_fwd_general_dh64(const acattn_problem& p, const acattn_fwd_out& o, bool full, hipStream_t stream);
int acattn_launch_fwd_general_dh128(const acattn_problem& p, const acattn_fwd_out& o, bool full, hipStream_t stream);

namespace {

__global__ void acattn_rng_kernel(int B, int nh, int L, uint64_t seed, float p_drop, float* noise, uint8_t* keep_after,
                                  uint8_t* keep_mask, uint8_t* keep_before) {
  const int groups = (L + 3) / 4;
  const size_t total = (size_t)B * nh * L * groups;
  for (size_t idx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const uint32_t row = (uint32_t)(idx / groups), grp = (uint32_t)(idx % groups);
    const RngGroup rg = rng_group(rng_key(seed), row, grp, p_drop);
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * grp + r;
      if (j >= L) break;
      const size_t o = (size_t)row * L + j;
      if (noise) noise[o] = rg.n[r];
      if (keep_after) keep_after[o] = (rg.keep_after >> r) & 1u;
      if (keep_mask) keep_mask[o] = (rg.keep_mask >> r) & 1u;
      if (keep_before) keep_before[o] = (rg.keep_before >> r) & 1u;
    }
  }
}

}  // namespace

int acattn_launch_fwd_fast(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream);
int acattn_launch_fwd_dma(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream);
int acattn_launch_fwd_stream(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream);

namespace {
// per host thread: a measurement / test hook must not race with launches another thread issues
thread_local int g_fwd_kernel = ACATTN_FWD_AUTO;
}
int acattn_fwd_kernel_choice(int which) {
  const int old = g_fwd_kernel;
  if (which >= ACATTN_FWD_AUTO && which <= ACATTN_FWD_GENERAL) g_fwd_kernel = which;
  return old;
}

int acattn_launch_fwd(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  // training hot paths (structured mask, counter RNG, gate, two_level), -100 = not applicable:
  //   streaming kernel (L <= 208; one wave per query block, operands straight from L2): the default, fastest at
  //   every measured shape (B = 512: L = 50 20.8 us against 21.8 staged and 33.3 general; L = 200, H = 64: 179 us
  //   against 536 general);
  //   LDS-staged kernels (L <= 64; LDS-DMA staging for 48 < L, register staging below): ACATTN_FWD_STAGED
  const int which = g_fwd_kernel;
  if (which != ACATTN_FWD_GENERAL) {
    if (which == ACATTN_FWD_AUTO || which == ACATTN_FWD_STREAM) {
      const int rc_stream = acattn_launch_fwd_stream(p, o, stream);
      if (rc_stream != -100) return rc_stream;
    }
    const int rc_dma = acattn_launch_fwd_dma(p, o, stream);
    if (rc_dma != -100) return rc_dma;
    const int rc_fast = acattn_launch_fwd_fast(p, o, stream);
    if (rc_fast != -100) return rc_fast;
  }
  const bool full = p.adversarial && (!p.two_level || o.after_spatial || o.before_spatial || o.perturbed_attention ||
                                      o.calibrated_attention);
  switch (p.H / p.n_heads) {
    case 16: return acattn_launch_fwd_general_dh16(p, o, full, stream);
    case 32: return acattn_launch_fwd_general_dh32(p, o, full, stream);
    case 64: return acattn_launch_fwd_general_dh64(p, o, full, stream);
    case 128: return acattn_launch_fwd_general_dh128(p, o, full, stream);
  }
  return -1;
}

int acattn_launch_rng(int B, int nh, int L, uint64_t seed, float p_drop, float* noise, uint8_t* keep_after,
                      uint8_t* keep_mask, uint8_t* keep_before, hipStream_t stream) {
  hipLaunchKernelGGL(acattn_rng_kernel, dim3(1024), dim3(256), 0, stream, B, nh, L, seed, p_drop, noise, keep_after,
                     keep_mask, keep_before);
  return (int)hipGetLastError();
}
