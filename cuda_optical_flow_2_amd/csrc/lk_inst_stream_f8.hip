// One family of instantiations of the templates in lk_launch.h (see there): the stream tick with eight columns per lane.
#include "lk_launch.h"

namespace ofx_launch {

int stream_lk_float_w8(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
    return launch_stream_mode_w8<OFX_MODE_LK_FLOAT, false>(radius, lv, n, S, stage_blocks, lds, st);
}

int levels_lk_float_w8(int radius, const LkLevelIn *lv, int n, hipStream_t st) { return launch_iter_mode_w8<OFX_MODE_LK_FLOAT, false, 0>(radius, lv, n, st); }

} // namespace ofx_launch
