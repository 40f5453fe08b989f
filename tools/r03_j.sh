# rows fetched two steps ahead through LDS (buffer_load ... lds, vmcnt(6)) against the one-step-ahead fetch into VGPRs
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k
mkdir -p $O
timeout -k 5 60 ./tools/ubench/lds_dma > $O/lds_dma.txt 2>&1; cat $O/lds_dma.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1
echo "pytest rc=$?"; tail -4 $O/tests.log
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
run dma A=1
run nodma OFX_LIB=libofx_nodma.so
run old OFX_LIB=libofx_old.so
run dma2 A=1
run nodma2 OFX_LIB=libofx_nodma.so
EXTRA="--workload 8k" run dma_8k A=1
EXTRA="--workload 8k" run nodma_8k OFX_LIB=libofx_nodma.so
EXTRA="--workload 1080p" run dma_1080p A=1
EXTRA="--workload 1080p" run nodma_1080p OFX_LIB=libofx_nodma.so
EXTRA="--mode compat_cpu" run dma_compat A=1
EXTRA="--mode compat_cpu" run nodma_compat OFX_LIB=libofx_nodma.so
