// general backward kernel, head size 32 (see acattn_bwd_general.inc)
#define ACATTN_BWD_DH 32
#include "acattn_bwd_general.inc"
