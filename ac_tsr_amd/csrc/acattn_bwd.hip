// Dispatch of acattn_calibrated_attention_bwd: streaming kernels, the row-resident kernel, then the general kernel
// (acattn_bwd_general.inc, one translation unit per head size: acattn_bwd_dh16.hip ... acattn_bwd_dh128.hip).
#include <stdlib.h>

#include "acattn_common.h"

int acattn_launch_bwd_general_dh16(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_general_dh32(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_general_dh64(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_general_dh128(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);

int acattn_launch_bwd_fast(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_stream(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_onerow(const acattn_problem& p, const acattn_bwd_io& io, bool accumulate, hipStream_t stream);

namespace {
// per host thread: a measurement / test hook must not race with launches another thread issues
thread_local int g_bwd_kernel = ACATTN_BWD_AUTO;
}
int acattn_bwd_kernel_choice(int which) {
  const int old = g_bwd_kernel;
  if (which >= ACATTN_BWD_AUTO && which <= ACATTN_BWD_ROW) g_bwd_kernel = which;
  return old;
}

bool acattn_bwd_onerow_applies(const acattn_problem& p, const acattn_bwd_io& io);

namespace {
bool onerow_split_applies(const acattn_problem& p, const acattn_bwd_io& io) {
  static const bool split = getenv("ACATTN_ONEROW_SPLIT") ? atoi(getenv("ACATTN_ONEROW_SPLIT")) != 0 : true;
  if (!(split && (io.d_attack_mask || io.d_penalty_part) && io.read_rows && io.n_read_rows == 1 && !io.active_qblocks && !io.attack_only && p.L <= 64))
    return false;
  const int dh = p.n_heads > 0 ? p.H / p.n_heads : 0;
  if (dh != 16 && dh != 32 && dh != 64) return false;  // the mask-only launch is the row-resident kernel's (acattn_launch_bwd_fast)
  acattn_bwd_io row_io = io;
  row_io.d_attack_mask = nullptr;
  row_io.d_penalty_part = nullptr;
  return acattn_bwd_onerow_applies(p, row_io);
}
}  // namespace

// acattn_bwd_io.dgate_summed: will the launch write the head-summed gate gradient?  (the one-row form, alone or behind the
// mask-only launch of the split)
bool acattn_bwd_gate_summed(const acattn_problem& p, const acattn_bwd_io& io) {
  if (g_bwd_kernel != ACATTN_BWD_AUTO || !io.dgate_logits) return false;
  return acattn_bwd_onerow_applies(p, io) || onerow_split_applies(p, io);
}

bool acattn_bwd_stream_pair_applies(const acattn_problem& p, const acattn_bwd_io& io);
bool acattn_bwd_pair_supported(const acattn_problem& p, const acattn_bwd_io& io) {
  // the streaming pair is what runs for L > 64 (or when pinned); L <= 64 takes the row-resident kernel, which has no second set
  const bool streaming = g_bwd_kernel == ACATTN_BWD_STREAM || (g_bwd_kernel == ACATTN_BWD_AUTO && p.L > 64);
  return streaming && acattn_bwd_stream_pair_applies(p, io);
}

int acattn_launch_bwd(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  if ((io.dqa2 || io.dka2 || io.d_ctx_calibrated2 || io.d_penalty_part2) && !acattn_bwd_pair_supported(p, io)) {
    acattn_set_error("attention backward: a second cotangent set was given but this launch cannot evaluate it "
                     "(ask acattn_calibrated_attention_bwd_pair_supported first)");
    return -1;
  }
  if (io.dgate_summed && !acattn_bwd_gate_summed(p, io)) {
    acattn_set_error("attention backward: dgate_summed requested but this launch does not take the one-row form "
                     "(ask acattn_calibrated_attention_bwd_gate_summed first)");
    return -1;
  }
  // training hot paths (structured mask, counter RNG, gate, two_level), -100 = not applicable:
  //   L <= 64: the row-resident kernel (acattn_bwd_fast.hip, one recomputation, a query block's row in registers).
  //     Inside the training step at B = 512, L = 50 its four calls take 71 / 108 / 103 / 57 us against 80 / 112 / 110 /
  //     92 us of the streaming pair (same box, same step: 1.76 against 1.81 ms per step); the attack-only call
  //     (dqa, dka alone) gains most.  In isolation with all three cotangents the order was the other way round
  //     (57 + 41 against 131 us) before the common random-number / mask code got cheaper for both.
  //   L > 64, or ACATTN_BWD_STREAM: the streaming two-kernel backward (acattn_bwd_stream.hip; needs io.workspace):
  //     L = 200 (H = 128, 4 heads) 0.87 ms against 7.0 ms of the general kernel.
  const int which = g_bwd_kernel;
  if (which == ACATTN_BWD_AUTO) {  // one position per sequence carries a context cotangent: one row of the backward (any L)
    const int rc_one = acattn_launch_bwd_onerow(p, io, false, stream);
    if (rc_one != -100) return rc_one;
    // ... and the attack mask a cotangent in every row (the attacked-loss pass through the last layer).  The backward is
    // linear in its cotangents: the mask cotangent alone (no context cotangent anywhere: every query block takes the
    // mask-only path of the row-resident kernel, which then owes dqa and dka only) + the read row's chain added on top.
    static const bool split = getenv("ACATTN_ONEROW_SPLIT") ? atoi(getenv("ACATTN_ONEROW_SPLIT")) != 0 : true;
    if (split && (io.d_attack_mask || io.d_penalty_part) && io.read_rows && io.n_read_rows == 1 && !io.active_qblocks && !io.attack_only && p.L <= 64) {
      acattn_bwd_io mask_io = io;
      mask_io.d_ctx_attacked = mask_io.d_ctx_calibrated = nullptr;
      mask_io.n_read_rows = 0;   // no block holds a read position
      mask_io.attack_only = 1;   // dqa, dka alone: everything else is written by the one-row launch
      const int rc_mask = acattn_launch_bwd_fast(p, mask_io, stream);
      if (rc_mask == 0) {
        acattn_bwd_io row_io = io;
        row_io.d_attack_mask = nullptr;
        row_io.d_penalty_part = nullptr;
        const int rc_row = acattn_launch_bwd_onerow(p, row_io, true, stream);
        if (rc_row != -100) return rc_row;
        // the mask launch ran: the read row's chain must not be dropped silently
        acattn_set_error("attention backward: the mask-only launch ran but the one-row kernel does not cover this problem (read-row chain not computed)");
        return -1;
      }
      if (rc_mask != -100) return rc_mask;
    }
    if (io.dgate_summed) {  // (unreachable while acattn_bwd_gate_summed mirrors the two one-row forms above)
      acattn_set_error("attention backward: dgate_summed was accepted but no one-row form ran");
      return -1;
    }
  }
  // (ACATTN_BWD_SHORT_STREAM=1, measurement: the streaming pair for L <= 64 as well)
  static const bool short_stream = getenv("ACATTN_BWD_SHORT_STREAM") && atoi(getenv("ACATTN_BWD_SHORT_STREAM")) != 0;
  const bool short_rows = p.L <= 64 && !short_stream;
  if (which == ACATTN_BWD_AUTO && short_rows) {
    const int rc_fast = acattn_launch_bwd_fast(p, io, stream);
    if (rc_fast != -100) return rc_fast;
  }
  if (which == ACATTN_BWD_AUTO || which == ACATTN_BWD_STREAM) {
    const int rc_stream = acattn_launch_bwd_stream(p, io, stream);
    if (rc_stream != -100) return rc_stream;
  }
  const int rc_fast = acattn_launch_bwd_fast(p, io, stream);
  if (rc_fast != -100) return rc_fast;
  switch (p.H / p.n_heads) {
    case 16: return acattn_launch_bwd_general_dh16(p, io, stream);
    case 32: return acattn_launch_bwd_general_dh32(p, io, stream);
    case 64: return acattn_launch_bwd_general_dh64(p, io, stream);
    case 128: return acattn_launch_bwd_general_dh128(p, io, stream);
  }
  return -1;
}
