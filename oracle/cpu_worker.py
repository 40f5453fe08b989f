"""One worker of bench.py's all-core CPU-baseline leg (test infrastructure, like everything under oracle/).

Started as a CHILD PROCESS by bench.py (never imports torch, never touches the GPU), pinned to one core:

    python oracle/cpu_worker.py <core> <width> <height> <levels> <window> <seconds> <use_ref 0|1>

Runs whole frame pairs -- both pyramids + every level, the reference's own cpu:: functions (oracle/_ref) or the C
restatement -- until <seconds> have passed (at least one pair) and prints "<pairs> <elapsed seconds>".
SURVEY.md section 8d(2): throughput mode = P independent pairs on P pinned processes.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    core, w, h, levels, window = (int(a) for a in sys.argv[1:6])
    seconds, use_ref = float(sys.argv[6]), sys.argv[7] == "1"
    try:
        os.sched_setaffinity(0, {core})
    except OSError:
        pass
    import oracle as orc
    from cuda_optical_flow_2_amd import synth

    p, n = synth.smooth_pair(w, h)
    p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
    run = (lambda: orc.Reference().flow_pair(p3, n3, levels)) if use_ref else (lambda: orc.Oracle().flow_pair(p3, n3, levels, window, "compat_cpu"))
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    sys.stdin.readline()   # all workers start together
    t0 = time.perf_counter()
    reps = 0
    while True:
        run()
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    print(reps, dt, flush=True)


if __name__ == "__main__":
    main()
