// streaming forward kernel, head size 32 (acattn_fwd_stream.inc)
#define ACATTN_STREAM_DH 32
#include "acattn_fwd_stream.inc"
