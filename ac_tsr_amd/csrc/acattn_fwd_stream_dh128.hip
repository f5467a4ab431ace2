// streaming forward kernel, head size 128 (see acattn_fwd_stream.inc)
#define ACATTN_STREAM_DH 128
#include "acattn_fwd_stream.inc"
