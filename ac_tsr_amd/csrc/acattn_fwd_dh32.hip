// general forward kernel, head size 32 (see acattn_fwd_general.inc)
#define ACATTN_FWD_DH 32
#include "acattn_fwd_general.inc"
