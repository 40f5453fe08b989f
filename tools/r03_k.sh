# deep fetch (rows through LDS, two steps ahead) chosen per launch: OFX_LK_DMA=0 / 1 on the same library, warm and cold rings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l
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
run dma0 OFX_LK_DMA=0
run dma1 OFX_LK_DMA=1
run dma0b OFX_LK_DMA=0
run dma1b OFX_LK_DMA=1
run cold_dma0 OFX_LK_DMA=0 OFX_BENCH_RING=40
run cold_dma1 OFX_LK_DMA=1 OFX_BENCH_RING=40
EXTRA="--workload 8k" run 8k_dma0 OFX_LK_DMA=0
EXTRA="--workload 8k" run 8k_dma1 OFX_LK_DMA=1
EXTRA="--workload 8k --batch 4" run 8k_b4_dma1 OFX_LK_DMA=1
EXTRA="--workload 1080p" run 1080p_dma0 OFX_LK_DMA=0
EXTRA="--workload 1080p" run 1080p_dma1 OFX_LK_DMA=1
EXTRA="--three-stage --batch 4" run b4_dma0 OFX_LK_DMA=0
EXTRA="--three-stage --batch 4" run b4_dma1 OFX_LK_DMA=1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream or sharded" > $O/tests_dma1.log 2>&1
echo "OFX_LK_DMA default pytest rc=$?"; tail -2 $O/tests_dma1.log
OFX_LK_DMA=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stream or sharded" > $O/tests_dma1f.log 2>&1
echo "OFX_LK_DMA=1 pytest rc=$?"; tail -2 $O/tests_dma1f.log
