// general backward kernel, head size 64 (see acattn_bwd_general.inc)
#define ACATTN_BWD_DH 64
#include "acattn_bwd_general.inc"
