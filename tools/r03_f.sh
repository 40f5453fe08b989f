# the LK march on buffer resources (scalar diet) against the old form: parity suite, then the launch time A/B on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1
echo "pytest rc=$?"; tail -5 $O/tests.log
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
run new A=1
run old OFX_LIB=libofx_old.so
run new2 A=1
run old2 OFX_LIB=libofx_old.so
EXTRA="--mode compat_cpu" run new_compat A=1
EXTRA="--mode compat_cpu" run old_compat OFX_LIB=libofx_old.so
EXTRA="--mode lk_float_fast" run new_fast A=1
EXTRA="--workload 8k" run new_8k A=1
EXTRA="--workload 8k" run old_8k OFX_LIB=libofx_old.so
EXTRA="--workload 1080p" run new_1080p A=1
EXTRA="--workload 1080p" run old_1080p OFX_LIB=libofx_old.so
