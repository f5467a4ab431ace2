// streaming forward kernel, head size 64 (acattn_fwd_stream.inc)
#define ACATTN_STREAM_DH 64
#include "acattn_fwd_stream.inc"
