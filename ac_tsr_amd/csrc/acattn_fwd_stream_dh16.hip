// streaming forward kernel, head size 16 (acattn_fwd_stream.inc)
#define ACATTN_STREAM_DH 16
#include "acattn_fwd_stream.inc"
