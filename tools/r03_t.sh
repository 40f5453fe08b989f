# full GPU suite + the iteration fuzzer after the row-window warp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3w
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > $O/tests.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -6 $O/tests.log
timeout -k 10 300 python tools/fuzz_iters.py 120 23 > $O/fuzz_iters.log 2>&1; echo "fuzz rc=$?"; tail -1 $O/fuzz_iters.log
