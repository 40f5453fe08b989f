# fused refinement iterations (lk_body_warp.h): parity, then A/B against the two-launch form; 3 vs 4 waves per SIMD
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "iter or refinement or literal" > $O/tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
B="python bench.py --no-cpu-baseline --no-extras"
run() { name=$1; shift
  env "$@" $B $EXTRA > $O/$name.json 2> $O/$name.err || echo "fail $name"
  python - $name $O/$name.json "$*" >> $O/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); r = d["roofline"]
    print(f"{sys.argv[1]:14s} {d['value']:10.1f} Mpix/s  frac {r['frac']:.4f}  check {d['self_check']}  {json.dumps(r.get('launches_us', d.get('launches_us', '')))}   [{sys.argv[3]}]")
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
  tail -1 $O/summary.txt
}
: > $O/summary.txt
EXTRA="--iters 5" run fused A=1
EXTRA="--iters 5" run unfused OFX_ITER_FUSED=0
EXTRA="--iters 5" run unfused_oldmarch OFX_ITER_FUSED=0 OFX_ITER_OLD_MARCH=1
EXTRA="--iters 5" run fused_b A=1
EXTRA="--iters 5 --workload 1080p" run 1080p_fused A=1
EXTRA="--iters 5 --workload 1080p" run 1080p_unfused OFX_ITER_FUSED=0
EXTRA="--iters 10 --workload 8k" run 8k_fused A=1
EXTRA="--iters 10 --workload 8k" run 8k_unfused OFX_ITER_FUSED=0
for n in fused unfused unfused_oldmarch 1080p_fused 1080p_unfused 8k_fused 8k_unfused; do python - $n $O/$n.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2])); print(sys.argv[1], d["roofline"].get("launches"))
PY
done
