# full GPU suite + the drop-in surface's timing with the staged transfers (and with them off)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
echo "pytest rc=$?"; tail -6 $O/tests.log
python - > $O/api.txt 2>&1 <<'PY'
import time, os, sys
sys.path.insert(0, ".")
import numpy as np
from cuda_optical_flow_2_amd import synth
from cuda_optical_flow_2_amd.compat import GpuCompat
gc = GpuCompat()
for nm, (w, h, L) in {"1080p": (1920, 1080, 4), "4k": (3840, 2160, 5)}.items():
    p, n = synth.smooth_pair(w, h)
    p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
    gc.flow_pair(p3, n3, L)
    t0 = time.perf_counter()
    for _ in range(3):
        gc.flow_pair(p3, n3, L)
    print(nm, "ms per pair", round((time.perf_counter() - t0) / 3 * 1e3, 2), "threads env", os.environ.get("OFX_STAGE_THREADS"))
PY
cat $O/api.txt
for t in -1 0 1 3 7; do OFX_STAGE_THREADS=$t python - >> $O/api_threads.txt 2>&1 <<'PY'
import time, os, sys
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import synth
from cuda_optical_flow_2_amd.compat import GpuCompat
gc = GpuCompat()
w, h, L = 3840, 2160, 5
p, n = synth.smooth_pair(w, h)
p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
gc.flow_pair(p3, n3, L)
t0 = time.perf_counter()
for _ in range(3):
    gc.flow_pair(p3, n3, L)
print("4k OFX_STAGE_THREADS", os.environ.get("OFX_STAGE_THREADS"), "ms per pair", round((time.perf_counter() - t0) / 3 * 1e3, 2))
PY
done
cat $O/api_threads.txt
