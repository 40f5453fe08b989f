# Samples power / clocks (rocm-smi) once a second while a bench.py configuration runs: is a launch power-bound?
#   gpurun -- 'bash tools/power_clock.sh gpurun_out/pc_i5 [bench.py args...]'
O=${1:?out prefix}; shift
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --no-extras --no-cpu-baseline --steps ${STEPS:-5000} "$@" > $O.json 2> $O.err &
BP=$!
sleep ${LEAD:-14}   # (import torch + session set-up: the timed region starts ~12 s in on a fresh box)
for i in $(seq 1 ${N:-10}); do
  echo "== sample $i $(date +%s.%N)" >> $O.smi
  timeout 5 rocm-smi --showpower --showclocks --showtemp --showuse 2>&1 | grep -i "power\|sclk\|mclk\|fclk\|socclk\|Temperature (Sensor junction)\|Temperature (Sensor memory)\|GPU use" >> $O.smi
  sleep 1
done
wait $BP
python tools/show_bench.py $O.json 2>&1 | head -8
