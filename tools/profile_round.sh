# Round evidence for profiles/: rocprofv3 kernel stats of the bench command (stream and plain path, iters = 5, compat_cpu) and
# HBM traffic per launch from separate --pmc passes (FETCH_SIZE / WRITE_SIZE; never combined with tracing).
#   gpurun -- 'bash tools/profile_round.sh r03'      then copy gpurun_out/prof_<tag>/summary/* into profiles/ and stamp the commit:
#   python tools/stamp_traffic.py   (the GPU box has no .git)
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O/summary
cd $R
stats() { # name, bench args...
  n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python bench.py --no-cpu-baseline --no-extras "$@" > $O/summary/${TAG}_bench4k_$n.json 2> $O/$n.err
  f=$(find $O/$n -name "*kernel_stats.csv" | head -1); cp "$f" $O/summary/${TAG}_bench4k_${n}_kernel_stats.csv
  echo "== $n"; head -4 "$f"
}
stats stream
stats plain --path plain
stats iters5 --iters 5 --steps 200
stats iters5_plain --iters 5 --steps 200 --path plain
stats compat_cpu --mode compat_cpu
stats random --frames random
# the other BASELINE configurations as written: file names say which
stats8() { n=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -- python bench.py --no-cpu-baseline --no-extras "$@" > $O/summary/${TAG}_bench_$n.json 2> $O/$n.err
  f=$(find $O/$n -name "*kernel_stats.csv" | head -1); cp "$f" $O/summary/${TAG}_bench_${n}_kernel_stats.csv
  echo "== $n"; head -4 "$f"
}
stats8 8k --workload 8k --steps 200
stats8 8k_iters10 --workload 8k --iters 10 --steps 12 --warmup 4
stats8 1080p --workload 1080p --steps 500
stats8 1080p_iters5 --workload 1080p --iters 5 --steps 50 --warmup 8
pmc() { # name, kernel substring, skip, pmc_run args...
  n=$1; k=$2; skip=$3; shift 3
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d $O/pmc_$n/$C -- python tools/pmc_run.py "$@" > /dev/null 2> $O/pmc_${n}_$C.err
  done
  python tools/pmc_parse.py $O/pmc_$n $k $skip | tee -a $O/summary/${TAG}_traffic_pmc.jsonl
}
pmc stream stream_kernel 3 4k stream lk_float
pmc plain lk_level_kernel 1 4k plain lk_float
pmc compat stream_kernel 3 4k stream compat_cpu
pmc iters5_lk lk_iter_kernel 1 4k plain lk_float 5
# traffic per launch for bench.py's roofline.traffic (keys: kernel, or kernel_<mode>_iters<n> for the non-default legs)
python - $O/summary/${TAG}_traffic_pmc.jsonl > $O/summary/traffic_latest.json <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip()]
names = ["stream_kernel", "lk_level_kernel", "stream_kernel_compat_cpu_iters1", "lk_level_kernel_lk_float_iters5"]
sys.path.insert(0, ".")
import bench
out = {"4k": {n: r.get("bytes") for n, r in zip(names, rows)},
       "kernel_source_sha16": bench.kernel_source_hash(), "measured_at_commit": None,
       "note": "HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB, rocprofv3 --pmc in separate passes (tools/profile_round.sh, tools/pmc_parse.py); "
               "lk_level_kernel_lk_float_iters5 is the mean over the five lk_iter_kernel launches of a pair (iteration 1 and three accumulating launches that also write the next warped image, one that does not)"}
print(json.dumps(out, indent=1))
PY
cat $O/summary/traffic_latest.json
# keep the merge small: drop the big traces
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O
