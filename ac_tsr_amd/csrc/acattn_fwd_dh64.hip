// general forward kernel, head size 64 (see acattn_fwd_general.inc)
#define ACATTN_FWD_DH 64
#include "acattn_fwd_general.inc"
