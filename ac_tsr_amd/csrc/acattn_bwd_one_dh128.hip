// general backward kernel, head size 128, two_level == 0 (see acattn_bwd_general.inc)
#define ACATTN_BWD_DH 128
#define ACATTN_BWD_ONE_LEVEL
#include "acattn_bwd_general.inc"
