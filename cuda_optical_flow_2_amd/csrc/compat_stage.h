// Chunked, multi-threaded host <-> device transfers for the host-pointer wrappers (compat_stage.cpp).
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace ofx_compat {
// Synchronous like the hipMemcpy they replace: at return the host source may be reused / the host destination is complete.
// Transfers below 3 MB (or all of them with OFX_STAGE_THREADS=-1) are one blocking hipMemcpy on the null stream.
int stage_h2d(void *d_dst, const void *h_src, size_t bytes);
int stage_d2h(void *h_dst, const void *d_src, size_t bytes);
// channel 0 of a tightly packed 3-channel w x h host image -> a 1-channel device plane of `pitch` bytes per row
int stage_h2d_ch0(uint8_t *d_dst1, int pitch, const uint8_t *h_src3, int w, int h);
int stage_threads(); // threads that move a transfer (the caller + the pool; OFX_STAGE_THREADS = pool size, default 3)
} // namespace ofx_compat
