// One family of instantiations of the templates in lk_launch.h (see there): the stream tick whose LK stage also writes the
// warped images of its pairs' second refinement iteration (lk_wave_buf's ITER = 3).
#include "lk_launch.h"

namespace ofx_launch {

int stream_lk_float_fast_wout(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
    return launch_stream_mode<OFX_MODE_LK_FLOAT, true, 3>(radius, lv, n, S, stage_blocks, lds, st);
}

} // namespace ofx_launch
