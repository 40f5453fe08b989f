# SQ + memory-side counters of one kernel for one configuration, separate rocprofv3 --pmc passes (no tracing alongside):
#   bash tools/pmc_kernel.sh <out dir under gpurun_out> <kernel substring> <pmc_run.py args...>
# prints, per counter, the mean over the launches of that kernel (first 3 skipped)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; K=$2; shift 2
rm -rf $O; mkdir -p $O
cd $R
for C in "GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCC_EA_WRREQ_STALL_sum TCC_EA_RDREQ_32B_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $O/$T -- python ${PMC_RUNNER:-tools/pmc_run.py} "$@" > /dev/null 2> $O/$T.err || echo "pass $T failed"
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
        print(f"{c:36s} {sum(l)/len(l):14.5g}   ({len(l)} launches)")
PY
find $O -name "*counter_collection.csv" -size +1M -delete
