// general backward kernel, head size 64, two_level == 0 (see acattn_bwd_general.inc)
#define ACATTN_BWD_DH 64
#define ACATTN_BWD_ONE_LEVEL
#include "acattn_bwd_general.inc"
