"""Experiment: aggregate throughput of two independent stream-pipeline sessions on two HIP streams (how much do
back-to-back dependent launches of one stream leave on the table?).  Usage: python tools/two_streams.py [n_sessions]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_optical_flow_2_amd import engine, synth

n_sess = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w, h, levels, window = 3840, 2160, 5, 9
dev = torch.device("cuda:0")
a, b = synth.smooth_pair(w, h, seed=1)
frames = [torch.from_numpy(x).to(dev) for x in (a, b)]
sessions = [engine.Session(w, h, levels, window, "lk_float") for _ in range(n_sess)]
streams = [torch.cuda.Stream() for _ in range(n_sess)]
for s in sessions:
    s.stream_begin()
def run(steps):
    for i in range(steps):
        for s, st in zip(sessions, streams):
            s.stream_submit(frames[i & 1], stream=st)
run(20)
torch.cuda.synchronize()
t0 = time.perf_counter()
steps = 200
run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{n_sess} sessions: {dt / steps * 1e6:.1f} us per round, {dt / (steps * n_sess) * 1e6:.1f} us per frame, "
      f"{w * h * steps * n_sess / dt / 1e6:.0f} Mpix/s aggregate")
