# sensitivity of the shipped launch to extra scalar / vector instructions per row step; the staged drop-in surface again
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3f
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras"
run() { name=$1; shift
  env "$@" $B > $O/$name.json 2> $O/$name.err || echo "fail $name"
  python - $name $O/$name.json "$*" >> $O/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print(f"{sys.argv[1]:14s} {d['value']:10.1f} Mpix/s  launch {r['avg_launch_us']:8.2f} us (min {r['min_launch_us']:.2f})  frac {r['frac']:.4f}  check {d['self_check']}   [{sys.argv[3]}]")
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
  tail -1 $O/summary.txt
}
: > $O/summary.txt
run base A=1
run salu40 OFX_LIB=libofx_salu40.so
run salu80 OFX_LIB=libofx_salu80.so
run valu40 OFX_LIB=libofx_valu40.so
run base2 A=1
for t in -1 1 3; do OFX_STAGE_THREADS=$t python - >> $O/api_threads.txt 2>&1 <<'PY'
import time, os, sys
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import synth
from cuda_optical_flow_2_amd.compat import GpuCompat
gc = GpuCompat()
for nm, (w, h, L) in {"1080p": (1920, 1080, 4), "4k": (3840, 2160, 5)}.items():
    fr = [synth.to_3ch(synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1]) for i in range(3)]
    loop = gc.frame_loop(w, h, L)
    loop.first(fr[0]); loop.step(fr[1])
    t0 = time.perf_counter()
    for i in range(6):
        loop.step(fr[(i + 2) % 3])
    print(nm, "OFX_STAGE_THREADS", os.environ.get("OFX_STAGE_THREADS"), "ms per frame", round((time.perf_counter() - t0) / 6 * 1e3, 2))
PY
done
grep -v amdgpu.ids $O/api_threads.txt
python -m pytest tests/test_gpu_surface.py -x -q -m gpu > $O/tests_surface.log 2>&1; tail -2 $O/tests_surface.log
