# full GPU suite, then the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3r
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - $O/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(d["value"], d["unit"], d["roofline"]["frac"], d["self_check"][:40])
for k, v in d.get("extra", {}).items():
    if isinstance(v, dict):
        r = v.get("roofline", {})
        print(f"  {k:32s} {v.get('value')}  frac {r.get('frac')}  as launched {r.get('frac_as_launched')}")
        for kk, vv in v.items():
            if isinstance(vv, dict) and "value" in vv:
                rr = vv.get("roofline", {})
                print(f"      {kk:28s} {vv.get('value')}  frac {rr.get('frac')}  as launched {rr.get('frac_as_launched')}")
PY
