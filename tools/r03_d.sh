# third batch: the drop-in surface with main.cu's buffer discipline (staged transfers on / off), the solve variants' cycle costs,
# and more frames per launch for the streamed iterations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e
mkdir -p $O
for t in -1 0 1 3 7; do OFX_STAGE_THREADS=$t python - >> $O/api_threads.txt 2>&1 <<'PY'
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
    for i in range(4):
        loop.step(fr[(i + 2) % 3])
    print(nm, "OFX_STAGE_THREADS", os.environ.get("OFX_STAGE_THREADS"), "ms per frame", round((time.perf_counter() - t0) / 4 * 1e3, 2))
PY
done
grep -v amdgpu.ids $O/api_threads.txt
./tools/ubench/solve_rates > $O/solve_rates.txt 2>&1; cat $O/solve_rates.txt
for b in 4 8; do python bench.py --iters 5 --batch $b --no-extras --no-cpu-baseline --steps 40 --warmup 4 > $O/iters5_b$b.json 2> $O/iters5_b$b.err; python -c "
import json; d=json.load(open('$O/iters5_b$b.json')); print('iters5 batch $b', d['value'], d['roofline']['frac'], d['roofline'].get('launches'))"; done
python -m pytest tests/test_gpu_surface.py -x -q -m gpu > $O/tests_surface.log 2>&1; tail -3 $O/tests_surface.log
