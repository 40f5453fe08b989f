// One family of instantiations of the templates in lk_launch.h (see there).
#include "lk_launch.h"

namespace ofx_launch {

int levels_lk_float_fast(int radius, const LkLevelIn *lv, int n, hipStream_t st)
{
    return launch_mode<OFX_MODE_LK_FLOAT, false, true>(radius, lv, n, st);
}

} // namespace ofx_launch
