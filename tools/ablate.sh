#!/bin/bash
# One parameterised ablation runner (replaces the one-off tools/r03_*.sh of round 3): every line of the plan is one bench.py run.
#   gpurun -- 'bash tools/ablate.sh gpurun_out/r4x tools/plans/<plan>.txt'        results: <outdir>/summary.txt, <outdir>/<name>.json
# Plan lines:   name | ENV=VALUE ENV=VALUE ... | bench.py arguments       ('#' starts a comment; empty fields allowed)
# Library variants built beforehand with OFX_BUILD_OUT / OFX_BUILD_DEFS are selected with OFX_LIB=<file> in the env field;
# ablated kernels (-DOFX_X_*) are wrong by construction: add OFX_BENCH_SKIP_CHECK=1.
O=${1:?outdir}; PLAN=${2:?plan file}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p $O
: > $O/summary.txt
grep -v '^\s*#' "$PLAN" | grep -v '^\s*$' | while IFS='|' read -r name envs args; do
  name=$(echo $name); 
  env $envs python bench.py --no-cpu-baseline --no-extras $args > $O/$name.json 2> $O/$name.err || echo "fail $name" | tee -a $O/summary.txt
  python - "$name" "$O/$name.json" "$envs $args" >> $O/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d["roofline"]
    wp = r.get("whole_pair", {})
    print(f"{sys.argv[1]:16s} {d['value']:10.1f} Mpix/s  {r.get('kind','?'):11s} launch {r['avg_launch_us']:8.2f} us (min {r['min_launch_us']:.2f}) frac {r['frac']:.4f}  pair {wp.get('kernel_us_per_pair','-')} us frac {wp.get('frac','-')}  check {d['self_check']}   [{sys.argv[3].strip()}]")
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
  tail -1 $O/summary.txt
done
