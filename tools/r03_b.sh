# second round-3 GPU batch: the repair tests, then the share of the pyramid and of the corner stage in the shipped launch
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "repaired or near_singular or leaves_the_patch or multi_frame_stream or sharded_stream or local_corner or beyond" > $O/tests.log 2>&1
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
run base A=1
run no_pyr OFX_LIB=libofx_exp.so OFX_STREAM_SKIP=1 OFX_BENCH_SKIP_CHECK=1
run no_corner OFX_LIB=libofx_exp.so OFX_STREAM_SKIP=2 OFX_BENCH_SKIP_CHECK=1
run lk_only OFX_LIB=libofx_exp.so OFX_STREAM_SKIP=3 OFX_BENCH_SKIP_CHECK=1
EXTRA=--three-stage run three_stage A=1
