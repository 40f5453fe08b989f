"""Tiny workload for rocprofv3 --pmc passes: a few level-0 LK launches at a BASELINE size (argv[1] = 4k|1080p|8k)."""
import sys
import torch
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import engine, synth
cfg = {"4k": (3840, 2160, 5, 9), "1080p": (1920, 1080, 4, 7), "8k": (7680, 4320, 6, 15)}[sys.argv[1] if len(sys.argv) > 1 else "4k"]
w, h, L, win = cfg
p, n = synth.smooth_pair(w, h)
s = engine.Session(w, h, L, win, "lk_float")
s.push_frame_host(p)
s.set_frame_host(n); s.build_pyramid()
for i in range(6):
    s.run_flow()
torch.cuda.synchronize()
s.close()
