"""Tiny workload for rocprofv3 --pmc passes at a BASELINE size.
    python tools/pmc_run.py [4k|1080p|8k] [plain|stream] [lk_float|lk_float_fast|compat_cpu] [iters]
plain: a few all-level lk_level_kernel launches of one pair; stream: a few stream_kernel launches (one per frame)."""
import sys
import torch
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import engine, synth
cfg = {"4k": (3840, 2160, 5, 9), "1080p": (1920, 1080, 4, 7), "8k": (7680, 4320, 6, 15)}[sys.argv[1] if len(sys.argv) > 1 else "4k"]
path = sys.argv[2] if len(sys.argv) > 2 else "plain"
mode = sys.argv[3] if len(sys.argv) > 3 else "lk_float"
iters = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 1
w, h, L, win = cfg
p, n = synth.smooth_pair(w, h)
import bench
TWO = path == "stream" and iters <= 1  # as bench.py runs it: borrowed frames from a ring of distinct buffers, two stages
B = engine.suggest_stream_batch(w, h, L, None, True, TWO)
if TWO and (B < 5 or B * w * h < 30e6):   # (bench.py's plan_stream: a short launch runs in three stages -- e.g. 8K, two frames per launch)
    TWO = False
    B = engine.suggest_stream_batch(w, h, L, None, True, False)
if iters > 1 and path == "stream":
    B = bench.iters_batch(w, h, L)   # (bench.py: pairs per launch of a stream with refinement iterations)
RING = (bench.cold_ring_size(B, TWO, w * h) if "--warm" not in sys.argv else bench.ring_size(B, TWO)) if path == "stream" else 0   # bench.py's ring: longer than the Infinity Cache
s = engine.Session(w, h, L, win, mode, stream_batch=B if path == "stream" else 1, borrow_frames=path == "stream", iters=iters, two_stage=TWO,
                   deep_fetch=bench.frames_hint(RING, w * h) if path == "stream" else 0)   # (as bench.py tells its sessions)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    if path == "plain":
        s.push_frame_host(p)
        s.set_frame_host(n); s.build_pyramid()
        for i in range(6):
            s.run_flow()
    else:
        ring = RING
        frames = [torch.from_numpy(synth.smooth_pair(w, h, 2.0 * (i % 4), 1.0 * (i % 4))[1]).cuda() for i in range(ring)]
        s.stream_begin()
        for i in range((10 if iters <= 1 else 6) * B):
            s.stream_submit(frames[i % ring])
torch.cuda.synchronize()
s.close()
print("pairs_per_launch", B if path == "stream" else 1)
