# cache-policy bits of the flow stores (buffer_store ... sc0 / nt / sc1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j
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
run nt A=1
run sc0_nt OFX_LIB=libofx_aux3.so
run nt_sc1 OFX_LIB=libofx_aux18.so
run sc1 OFX_LIB=libofx_aux16.so
run sc0 OFX_LIB=libofx_aux1.so
run nt2 A=1
