for v in ${VARIANTS:-hip nosolve nohbox nosh}; do
  OFX_LIB=libofx_$v.so OFX_BENCH_SKIP_CHECK=1 python bench.py --no-cpu-baseline --no-extras > gpurun_out/r2e_$v.json 2> gpurun_out/r2e_$v.err || echo "fail $v"
  python -c "
import json; d=json.load(open('gpurun_out/r2e_$v.json')); print('$v', d['value'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"
done
