// Dispatch of the streaming forward kernel (acattn_fwd_stream.inc; one translation unit per head size).
#include <stdlib.h>

#include "acattn_common.h"

int acattn_launch_fwd_stream_dh16(const acattn_problem& p, const acattn_fwd_out& o, int pre, hipStream_t stream);
int acattn_launch_fwd_stream_dh32(const acattn_problem& p, const acattn_fwd_out& o, int pre, hipStream_t stream);
int acattn_launch_fwd_stream_dh64(const acattn_problem& p, const acattn_fwd_out& o, int pre, hipStream_t stream);
int acattn_launch_fwd_stream_dh128(const acattn_problem& p, const acattn_fwd_out& o, int pre, hipStream_t stream);

namespace {
thread_local bool g_penalty_written = false;
}
void acattn_penalty_written_set(bool v) { g_penalty_written = v; }
bool acattn_penalty_written() { return g_penalty_written; }

// Returns -100 when the problem is outside this kernel's domain (the caller then tries the other kernels).
int acattn_launch_fwd_stream(const acattn_problem& p, const acattn_fwd_out& o, hipStream_t stream) {
  static const bool enabled = getenv("ACATTN_STREAM") ? atoi(getenv("ACATTN_STREAM")) != 0 : true;
  const bool ok = enabled && p.L <= 208 && (int64_t)p.B * p.n_heads * p.L * p.L < (1LL << 30) &&
                  (int64_t)p.B * p.L * p.H < (1LL << 30) && p.mask_mode == ACATTN_MASK_STRUCTURED &&
                  p.rng_mode == ACATTN_RNG_COUNTER && p.w_order && p.w_dist &&
                  (!p.adversarial || (p.combine_option == ACATTN_COMBINE_GATE && p.two_level)) && !o.after_spatial &&
                  !o.before_spatial && !o.perturbed_attention && !o.calibrated_attention;
  if (!ok) return -100;
  // the producer's affine planes and gate probabilities are used when BOTH are there (the spatial-only operator has
  // no gate); gate probabilities without the planes are left to the other kernels
  const bool gate_prob = p.adversarial && p.gate_is_prob;
  const int pre = p.affine && (gate_prob || !p.adversarial) ? 1 : 0;
  if (gate_prob && !pre) return -100;
  // (Round 3 sent dh = 64, L > 64 without the producer extras to the general kernel: that instantiation keeps 33 registers
  // in scratch, and two work-in-progress builds that spilled had produced wrong tiles.  Round 4 could not reproduce that
  // with any committed source: builds forced to spill 85-109 registers at 5 and 6 waves per SIMD are bit-identical to the
  // spill-free build at B = 128 / 256 / 512, every reload is dominated by its store (tools/scratch_dominance.py), and this
  // instantiation equals the oracle at 3,328 waves (tests/test_hip_onehop.py, id "dh64_L200_spilling") -- DESIGN 4.1.)
  switch (p.H / p.n_heads) {
    case 16: return acattn_launch_fwd_stream_dh16(p, o, pre, stream);
    case 32: return acattn_launch_fwd_stream_dh32(p, o, pre, stream);
    case 64: return acattn_launch_fwd_stream_dh64(p, o, pre, stream);
    case 128: return acattn_launch_fwd_stream_dh128(p, o, pre, stream);  // [r3] one wave per SIMD, spill-free
  }
  return -100;
}
