#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / duration of the accumulating launch that also warps (lk_iter_kernel<4,1,false,2>) for diagnostic library
# builds that leave one kind of load or store out (-DOFX_X_NO_OUTROWS / NO_OLDFLOW / NO_TAPS / NO_WSTORE: results wrong by
# construction): which of its memory operations the launch's fabric reads belong to.
#   gpurun -- 'bash tools/pmc_fetch_variants.sh gpurun_out/r4i libofx_hip.so libofx_x_NO_OUTROWS.so ...'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/$1; shift
mkdir -p $O; cd $R
for lib in "$@"; do
  n=${lib%.so}; n=${n#libofx_}
  for C in FETCH_SIZE WRITE_SIZE; do
    OFX_LIB=$lib rocprofv3 --pmc $C --output-format csv -d $O/$n/$C -- python tools/pmc_run.py 4k stream lk_float 5 > /dev/null 2> $O/${n}_$C.err
  done
  OFX_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n/stats -- python tools/pmc_run.py 4k stream lk_float 5 > /dev/null 2> $O/${n}_stats.err
  f=$(find $O/$n/stats -name "*kernel_stats.csv" | head -1)
  echo "== $n  $(python tools/pmc_parse.py $O/$n 'lk_iter_kernel<4, 1, false, 2,' 2)  avg_ns $(grep 'lk_iter_kernel<4, 1, false, 2,' $f | awk -F'","' '{print $4}' | tr -d '"')" | tee -a $O/summary.txt
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
