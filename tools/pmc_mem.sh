set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_warp_mem; rm -rf $O; mkdir -p $O
cd $R
for C in "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum" "TCC_EA_RD_UNCACHED_32B_sum TCC_EA_WRREQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  T=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d $O/$T -- python tools/pmc_run.py 4k plain lk_float 5 > /dev/null 2> $O/$T.err || echo "pass $T failed"
done
python - "$O" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
for k in ("warp_u8_kernel", "lk_level_kernel"):
  print("==", k)
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
        print(f"{c:36s} {sum(l)/len(l):14.4g}   ({len(l)} launches)")
PY
find $O -name "*counter_collection.csv" -size +1M -delete
