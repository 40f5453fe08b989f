# the fast bilateral filter: +-1 LSB test, then both kernels timed (frontend leg of bench.py in isolation)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_surface.py -x -q -m gpu > $O/tests.log 2>&1
echo "pytest rc=$?"; tail -12 $O/tests.log
python - > $O/frontend.txt 2>&1 <<'PY'
import sys, json
sys.path.insert(0, ".")
import bench
class A: pass
a = A(); a.gpus = 1; a.workload = "4k"; a.batch = 8; a.two_stage = True; a.frames = "texture"; a.path = "stream"; a.iters = 1; a.mode = "lk_float"
r = bench.Run(a)
print(json.dumps(r.leg_frontend(), indent=1))
PY
cat $O/frontend.txt | grep -v amdgpu.ids
# roctx ranges of the session's stages show up in a marker trace
cd /tmp && export TMPDIR=/tmp
OFX_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/roctx -- python $GRAFT_REPO_ROOT/bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 2 > $GRAFT_REPO_ROOT/$O/roctx_bench.json 2> $GRAFT_REPO_ROOT/$O/roctx.err
cd $GRAFT_REPO_ROOT
f=$(find $O/roctx -name "*marker_api_trace.csv" | head -1); echo "marker file: $f"; head -3 "$f"; cut -d, -f3 "$f" | sort | uniq -c | sort -rn | head -5
find $O/roctx -name "*kernel_trace.csv" -delete
