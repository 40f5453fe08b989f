# what the flow stores cost the shipped launch: all rows into one MB of L2 (no HBM write stream), no stores at all, with and without the solve
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h
mkdir -p $O
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
run tiny OFX_LIB=libofx_tiny.so OFX_BENCH_SKIP_CHECK=1
run nostore OFX_LIB=libofx_nostore.so OFX_BENCH_SKIP_CHECK=1
run nosolve OFX_LIB=libofx_nosolve.so OFX_BENCH_SKIP_CHECK=1
run tiny_nosolve OFX_LIB=libofx_tinyfast.so OFX_BENCH_SKIP_CHECK=1
run base2 A=1
