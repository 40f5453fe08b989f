# Round evidence for profiles/: rocprofv3 kernel stats of the bench command (the headline = config as written, iters = 1, plain path, compat_cpu, ...) and
# HBM traffic per launch from separate --pmc passes (FETCH_SIZE / WRITE_SIZE; never combined with tracing) -- taken FIRST, so that the bench
# lines of the profiled runs carry a traffic figure measured on the same sources.
#   gpurun -- 'bash tools/profile_round.sh r04'      then copy gpurun_out/prof_<tag>/summary/* into profiles/ and stamp the commit:
#   python tools/stamp_traffic.py   (the GPU box has no .git)
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
# PROFILE_ONLY="name name ...": only those steps (pmc: stream iter2 iter1 plain compat; stats: headline iters1 ... 1080p_iters1), into the
# directory as it stands -- to finish a round that a failing step cut short
want() { [ -z "$PROFILE_ONLY" ] || [[ " $PROFILE_ONLY " == *" $1 "* ]]; }
[ -n "$PROFILE_ONLY" ] || rm -rf $O
mkdir -p $O/summary
cd $R
pmc() { # name, kernel substring, skip, pmc_run args...
  n=$1; k=$2; skip=$3; shift 3
  want $n || return 0
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$n/$C -- python tools/pmc_run.py "$@" > $O/pmc_${n}_$C.out 2> $O/pmc_${n}_$C.err
  done
  python tools/pmc_parse.py $O/pmc_$n "$k" $skip | tee -a $O/summary/${TAG}_traffic_pmc.jsonl
}
pmc stream stream_kernel 3 4k stream lk_float
pmc iter2 "lk_iter_kernel<4, 1, false, 2," 2 4k stream lk_float 5
pmc iter1 "lk_iter_kernel<4, 1, false, 1," 1 4k stream lk_float 5
pmc plain lk_level_kernel 1 4k plain lk_float
pmc compat stream_kernel 3 4k stream compat_cpu
# traffic per launch for bench.py's roofline.traffic (keys: bench.py line(): tkey)
if want stream; then
python - $O/summary/${TAG}_traffic_pmc.jsonl $O > $O/summary/traffic_latest.json <<'PY'
import json, re, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
names = ["stream_kernel", "lk_iter_kernel_iter2", "lk_iter_kernel_iter1", "lk_level_kernel", "stream_kernel_compat_cpu"]
runs = ["stream", "iter2", "iter1", "plain", "compat"]
ppl = {}
for n, r in zip(names, runs):
    try:
        m = re.search(r"pairs_per_launch (\d+)", open(f"{sys.argv[2]}/pmc_{r}_FETCH_SIZE.out").read())
        ppl[n] = int(m.group(1)) if m else None
    except OSError:
        ppl[n] = None
sys.path.insert(0, ".")
import bench
out = {"4k": {n: r.get("bytes") for n, r in zip(names, rows)}, "pairs_per_launch": ppl,
       "kernel_source_sha16": bench.kernel_source_hash(), "measured_at_commit": None,
       "note": "HBM-side bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, rocprofv3 --pmc in separate passes (tools/profile_round.sh, tools/pmc_parse.py; "
               "the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE tallies 128-byte requests at 64); inputs from a ring longer than the Infinity "
               "Cache, as bench.py's; lk_iter_kernel_iter2 = an accumulating launch that also writes the next warped image (8 pairs per launch), _iter1 = the last iteration"}
print(json.dumps(out, indent=1))
PY
cat $O/summary/traffic_latest.json
# the bench lines below carry this figure: bench.py reads profiles/traffic_latest.json and checks its kernel_source_sha16 against the sources
cp $O/summary/traffic_latest.json $R/profiles/traffic_latest.json
fi
stats() { # name, bench args...
  n=$1; shift
  want $n || return 0
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python bench.py --no-cpu-baseline --no-extras "$@" > $O/summary/${TAG}_bench4k_$n.json 2> $O/$n.err
  f=$(find $O/$n -name "*kernel_stats.csv" | head -1); cp "$f" $O/summary/${TAG}_bench4k_${n}_kernel_stats.csv
  echo "== $n"; head -4 "$f"
}
stats headline
stats iters1 --iters 1
stats iters1_warm --iters 1 --ring 20
stats plain --iters 1 --path plain
stats iters5_plain --path plain --steps 200
stats compat_cpu --iters 1 --mode compat_cpu
stats random --iters 1 --frames random
# the other BASELINE configurations as written: file names say which
stats8() { n=$1; shift
  want $n || return 0
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python bench.py --no-cpu-baseline --no-extras "$@" > $O/summary/${TAG}_bench_$n.json 2> $O/$n.err
  f=$(find $O/$n -name "*kernel_stats.csv" | head -1); cp "$f" $O/summary/${TAG}_bench_${n}_kernel_stats.csv
  echo "== $n"; head -4 "$f"
}
stats8 8k_iters10 --workload 8k --steps 12 --warmup 4
stats8 8k_iters1 --workload 8k --iters 1 --steps 200
stats8 1080p_iters5 --workload 1080p --steps 50 --warmup 8
stats8 1080p_iters1 --workload 1080p --iters 1 --steps 500
# keep the merge small: drop the big traces
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O
