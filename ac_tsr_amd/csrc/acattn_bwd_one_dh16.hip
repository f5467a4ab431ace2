// general backward kernel, head size 16, two_level == 0 (see acattn_bwd_general.inc)
#define ACATTN_BWD_DH 16
#define ACATTN_BWD_ONE_LEVEL
#include "acattn_bwd_general.inc"
