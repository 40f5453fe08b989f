// One family of instantiations of the templates in lk_launch.h (see there).
#include "lk_launch.h"

namespace ofx_launch {

int stream_compat_cpu(int radius, const LkLevelIn *lv, int n, StreamArgs &S, const int *stage_blocks, size_t lds, hipStream_t st)
{
    return launch_stream_mode<OFX_MODE_COMPAT_CPU, false>(radius, lv, n, S, stage_blocks, lds, st);
}

} // namespace ofx_launch
