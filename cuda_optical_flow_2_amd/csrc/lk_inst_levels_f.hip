// One family of instantiations of the templates in lk_launch.h (see there).
#include "lk_launch.h"

namespace ofx_launch {

int levels_lk_float(int radius, const LkLevelIn *lv, int n, bool sums, hipStream_t st)
{
    return sums ? launch_mode<OFX_MODE_LK_FLOAT, true, false>(radius, lv, n, st) : launch_mode<OFX_MODE_LK_FLOAT, false, false>(radius, lv, n, st);
}

} // namespace ofx_launch
