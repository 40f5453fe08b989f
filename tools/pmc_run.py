"""Tiny workload for rocprofv3 --pmc passes at a BASELINE size.
    python tools/pmc_run.py [4k|1080p|8k] [plain|stream]
plain: a few all-level lk_level_kernel launches of one pair; stream: a few stream_kernel launches (one per frame)."""
import sys
import torch
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import engine, synth
cfg = {"4k": (3840, 2160, 5, 9), "1080p": (1920, 1080, 4, 7), "8k": (7680, 4320, 6, 15)}[sys.argv[1] if len(sys.argv) > 1 else "4k"]
path = sys.argv[2] if len(sys.argv) > 2 else "plain"
w, h, L, win = cfg
p, n = synth.smooth_pair(w, h)
s = engine.Session(w, h, L, win, "lk_float", stream_batch=4 if path == "stream" else 1)  # as bench.py runs it
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    if path == "plain":
        s.push_frame_host(p)
        s.set_frame_host(n); s.build_pyramid()
        for i in range(6):
            s.run_flow()
    else:
        frames = [torch.from_numpy(x).cuda() for x in (p, n)]
        s.stream_begin()
        for i in range(40):
            s.stream_submit(frames[i & 1])
torch.cuda.synchronize()
s.close()
