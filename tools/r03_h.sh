# lockstep box sums (hbox4x5) against the per-quantity chains, same box; parity of the windows first
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/tests.log 2>&1
echo "pytest rc=$?"; tail -3 $O/tests.log
B="python bench.py --no-cpu-baseline --no-extras"
run() { name=$1; shift
  env "$@" $B $EXTRA > $O/$name.json 2> $O/$name.err || echo "fail $name"
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
run lock A=1
run nolock OFX_LIB=libofx_nolock.so
run old OFX_LIB=libofx_old.so
run lock2 A=1
run nolock2 OFX_LIB=libofx_nolock.so
EXTRA="--workload 8k" run lock_8k A=1
EXTRA="--workload 8k" run nolock_8k OFX_LIB=libofx_nolock.so
EXTRA="--workload 1080p" run lock_1080p A=1
EXTRA="--workload 1080p" run nolock_1080p OFX_LIB=libofx_nolock.so
