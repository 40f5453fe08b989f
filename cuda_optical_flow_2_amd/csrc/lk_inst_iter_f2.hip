// One family of instantiations of the templates in lk_launch.h (see there): refinement iterations on the buffer march, ITER = 2.
#include "lk_launch.h"

namespace ofx_launch {

int iter2_lk_float(int radius, const LkLevelIn *lv, int n, hipStream_t st) { return launch_iter_mode<OFX_MODE_LK_FLOAT, false, 2>(radius, lv, n, st); }

} // namespace ofx_launch
