# refinement iterations: ITER = 2 at 3 vs 4 waves per SIMD (libofx_w4.so: launch bounds 4, a few spills in edge tiles)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras"
run() { name=$1; shift
  env "$@" $B $EXTRA > $O/$name.json 2> $O/$name.err || echo "fail $name"
  python - $name $O/$name.json "$*" >> $O/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print(f"{sys.argv[1]:14s} {d['value']:10.1f} Mpix/s  frac {r['frac']:.4f} as launched {r.get('frac_as_launched')}  check {d['self_check'][:2]}  {json.dumps({k: v['avg_us'] for k, v in r.get('launches', {}).items()})}   [{sys.argv[3]}]")
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
  tail -1 $O/summary.txt
}
: > $O/summary.txt
EXTRA="--iters 5" run w3 A=1
EXTRA="--iters 5" run w4 OFX_LIB=libofx_w4.so
EXTRA="--iters 5 --workload 1080p" run 1080p_w3 A=1
EXTRA="--iters 5 --workload 1080p" run 1080p_w4 OFX_LIB=libofx_w4.so
EXTRA="--iters 10 --workload 8k" run 8k_w3 A=1
EXTRA="--iters 10 --workload 8k" run 8k_w4 OFX_LIB=libofx_w4.so
EXTRA="--iters 5 --batch 8" run b8_w3 A=1
EXTRA="--iters 5 --batch 2" run b2_w3 A=1
