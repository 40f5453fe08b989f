// One family of instantiations of the templates in lk_launch.h (see there).
#include "lk_launch.h"

namespace ofx_launch {

int levels_compat_cpu(int radius, const LkLevelIn *lv, int n, bool sums, hipStream_t st)
{
    return sums ? launch_mode<OFX_MODE_COMPAT_CPU, true, false>(radius, lv, n, st) : launch_mode<OFX_MODE_COMPAT_CPU, false, false>(radius, lv, n, st);
}

} // namespace ofx_launch
