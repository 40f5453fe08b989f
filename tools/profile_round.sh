set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_l
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stream -- python bench.py --no-cpu-baseline --no-extras > $O/bench_stream.json 2> $O/stream.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/plain -- python bench.py --no-cpu-baseline --no-extras --path plain > $O/bench_plain.json 2> $O/plain.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_stream -- python tools/pmc_run.py 4k stream > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_stream -- python tools/pmc_run.py 4k stream > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_plain -- python tools/pmc_run.py 4k plain > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_plain -- python tools/pmc_run.py 4k plain > /dev/null 2>&1
mkdir -p $O/pmc_stream $O/pmc_plain
cp -r $O/pmc_fetch_stream $O/pmc_write_stream $O/pmc_stream/
cp -r $O/pmc_fetch_plain $O/pmc_write_plain $O/pmc_plain/
python tools/pmc_parse.py $O/pmc_stream stream_kernel 3 > $O/traffic_stream.json
python tools/pmc_parse.py $O/pmc_plain lk_level_kernel 1 > $O/traffic_plain.json
cat $O/traffic_stream.json $O/traffic_plain.json
find $O -name "*kernel_stats.csv" | head
# keep the merge small: drop the big traces
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +2M -delete
du -sh $O
