// general forward kernel, head size 16 (see acattn_fwd_general.inc)
#define ACATTN_FWD_DH 16
#include "acattn_fwd_general.inc"
