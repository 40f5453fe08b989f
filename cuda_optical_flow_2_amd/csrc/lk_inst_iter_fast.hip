// One family of instantiations of the templates in lk_launch.h (see there): refinement iterations on the buffer march.
#include "lk_launch.h"

namespace ofx_launch {

int iter_lk_float_fast(int radius, const LkLevelIn *lv, int n, int iter, hipStream_t st)
{
    if (iter == 3) return launch_iter_mode<OFX_MODE_LK_FLOAT, true, 3>(radius, lv, n, st);
    return iter == 2 ? launch_iter_mode<OFX_MODE_LK_FLOAT, true, 2>(radius, lv, n, st) : launch_iter_mode<OFX_MODE_LK_FLOAT, true, 1>(radius, lv, n, st);
}

} // namespace ofx_launch
