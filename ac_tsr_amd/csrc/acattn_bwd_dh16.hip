// general backward kernel, head size 16 (see acattn_bwd_general.inc)
#define ACATTN_BWD_DH 16
#include "acattn_bwd_general.inc"
