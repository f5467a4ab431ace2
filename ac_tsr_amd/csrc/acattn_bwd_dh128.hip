// general backward kernel, head size 128 (see acattn_bwd_general.inc)
#define ACATTN_BWD_DH 128
#include "acattn_bwd_general.inc"
