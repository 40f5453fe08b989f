# SQ counters of one kernel for one configuration, separate rocprofv3 --pmc passes (no tracing alongside):
#   bash tools/pmc_counters.sh <out dir under gpurun_out> <kernel substring> <pmc_run.py args...>
#   PMC_PROG=tools/bil_bench.py PMC_LDS=1 bash tools/pmc_counters.sh ...   another program under the counters; the LDS counters too
# prints, per counter, the mean over the launches of that kernel (first 3 skipped)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; K=$2; shift 2
rm -rf $O; mkdir -p $O
cd $R
PROG=${PMC_PROG:-tools/pmc_run.py}
EXTRA=()
if [ -n "$PMC_LDS" ]; then EXTRA=("SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"); fi
SETS=("${EXTRA[@]}" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "GRBM_GUI_ACTIVE")
# PMC_MEM=1: the memory side instead -- what the L2s ask of the fabric, how long the requests stay out, who stalls whom
if [ -n "$PMC_MEM" ]; then SETS=("GRBM_GUI_ACTIVE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum" "TCC_TAG_STALL_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum"); fi
# (a pass with TA_BUSY_avr / TA_*_STALLED_BY_TC_CYCLES_sum never returned on this pool: left out; every pass has its own time limit)
for C in "${SETS[@]}"; do
  T=$(echo $C | tr ' ' '_')
  timeout -k 10 ${PMC_PASS_LIMIT:-150} rocprofv3 --pmc $C --output-format csv -d $O/$T -- python $PROG "$@" > /dev/null 2> $O/$T.err || echo "pass $T failed"
  echo "pass $T done" >> $O/progress.txt
done
python - "$O" "$K" <<'PY'
import csv, glob, os, sys
root, k = sys.argv[1], sys.argv[2]
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    per = {}
    for row in csv.DictReader(open(f)):
        if k not in row.get("Kernel_Name", ""): continue
        key = (row["Counter_Name"], int(row["Dispatch_Id"]))
        per[key] = per.get(key, 0.0) + float(row["Counter_Value"])
    by = {}
    for (c, d), v in sorted(per.items(), key=lambda kv: kv[0][1]): by.setdefault(c, []).append(v)
    for c, l in by.items():
        l = l[3:] if len(l) > 3 else l
        print(f"{c:24s} {sum(l)/len(l):14.4g}   ({len(l)} launches)")
PY
find $O -name "*counter_collection.csv" -size +1M -delete
