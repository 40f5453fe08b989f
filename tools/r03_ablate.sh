# Round-3 ablation of the default stream launch (4K, eight pairs per launch, two stages): what each stage of the LK march costs,
# what the launch costs with nothing but its loads, LDS exchange and streaming stores (OFX_X_SKELETON), the age skew of the
# strips, and the per-instruction VALU rates.  Library variants are built beforehand (OFX_BUILD_OUT / OFX_BUILD_DEFS).
#   gpurun -- 'bash tools/r03_ablate.sh'          results: gpurun_out/r3a/summary.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3a
mkdir -p $O
B="python bench.py --no-cpu-baseline --no-extras"
run() { # name, env assignments...
  name=$1; shift
  env "$@" $B > $O/$name.json 2> $O/$name.err || echo "fail $name" | tee -a $O/summary.txt
  python - $name $O/$name.json "$*" >> $O/summary.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    r = d["roofline"]
    print(f"{sys.argv[1]:14s} {d['value']:10.1f} Mpix/s  launch {r['avg_launch_us']:8.2f} us (min {r['min_launch_us']:.2f})  frac {r['frac']:.4f}  check {d['self_check']}   [{sys.argv[3]}]")
except Exception as e:
    print(sys.argv[1], "no result:", e)
PY
  tail -1 $O/summary.txt
}
: > $O/summary.txt
run base A=1
run skel OFX_LIB=libofx_skel.so OFX_BENCH_SKIP_CHECK=1
run nosolve OFX_LIB=libofx_nosolve.so OFX_BENCH_SKIP_CHECK=1
run nohbox OFX_LIB=libofx_nohbox.so OFX_BENCH_SKIP_CHECK=1
run neither OFX_LIB=libofx_neither.so OFX_BENCH_SKIP_CHECK=1
run lk_only OFX_LIB=libofx_exp.so OFX_STREAM_SKIP=3 OFX_BENCH_SKIP_CHECK=1
run no_lk OFX_LIB=libofx_exp.so OFX_STREAM_SKIP=8 OFX_BENCH_SKIP_CHECK=1
run skel_lk_only OFX_LIB=libofx_skelexp.so OFX_STREAM_SKIP=3 OFX_BENCH_SKIP_CHECK=1
run skew_a OFX_LK_SKEW=104,101,99,96
run skew_b OFX_LK_SKEW=108,103,97,92
run skew_c OFX_LK_SKEW=112,104,96,88
run skew_d OFX_LK_SKEW=96,99,101,104
run base2 A=1
python tools/stream_timeline.py 8 > $O/timeline.txt 2>&1 || echo "timeline failed"
OFX_LK_SKEW=108,103,97,92 python tools/stream_timeline.py 8 > $O/timeline_skew_b.txt 2>&1 || echo "timeline failed"
./tools/ubench/valu_rates > $O/valu_rates.txt 2>&1 || echo "valu_rates failed"
cat $O/summary.txt
