// general forward kernel, head size 128 (see acattn_fwd_general.inc)
#define ACATTN_FWD_DH 128
#include "acattn_fwd_general.inc"
