// placeholder until the backward kernel lands
#include "acattn_common.h"
int acattn_launch_bwd(const acattn_problem& p, const acattn_bwd_io& io, hipStream_t stream) {
  acattn_set_error("backward kernel not built");
  return -2;
}
