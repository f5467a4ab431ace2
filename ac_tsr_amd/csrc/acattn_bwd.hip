// Dispatch of acattn_calibrated_attention_bwd: streaming kernels, the row-resident kernel, then the general kernel
// (acattn_bwd_general.inc, one translation unit per head size: acattn_bwd_dh16.hip ... acattn_bwd_dh128.hip).
#include "acattn_common.h"

int acattn_launch_bwd_general_dh16(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_general_dh32(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_general_dh64(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_general_dh128(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);

int acattn_launch_bwd_fast(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);
int acattn_launch_bwd_stream(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream);

namespace {
int g_bwd_kernel = ACATTN_BWD_AUTO;
}
int acattn_bwd_kernel_choice(int which) {
  const int old = g_bwd_kernel;
  if (which >= ACATTN_BWD_AUTO && which <= ACATTN_BWD_ROW) g_bwd_kernel = which;
  return old;
}

int acattn_launch_bwd(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  // training hot paths (structured mask, counter RNG, gate, two_level), -100 = not applicable:
  //   streaming two-kernel backward (acattn_bwd_stream.hip; needs io.workspace): the default.  B = 512, all three
  //   cotangents: L = 50 57 + 41 us against 131 us row-resident; L = 200 (H = 128, 4 heads) 0.87 ms against 7.0 ms;
  //   row-resident kernel, one recomputation (acattn_bwd_fast.hip, L <= 64): ACATTN_BWD_ROW, or no workspace
  const int which = g_bwd_kernel;
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
