import sys, time
import torch
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import engine, synth
w, h, L, win = 3840, 2160, 5, 9
frames = [torch.from_numpy(synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1]).cuda() for i in range(4)]
s = engine.Session(w, h, L, win, "lk_float")
s.set_frame_device(frames[0]); s.build_pyramid(); s.swap()
for i in range(20): s.submit_device(frames[(i + 1) % 4])
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for i in range(N): s.submit_device(frames[(i + 1) % 4])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/N:.1f} us/pair, total {1e6*(t2-t0)/N:.1f} us/pair (pipelined submit)")
# plain path on one stream
t0 = time.perf_counter()
for i in range(N):
    s.set_frame_device(frames[(i + 1) % 4]); s.build_pyramid(); s.run_flow(); s.swap()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e6*(t1-t0)/N:.1f} us/pair, total {1e6*(t2-t0)/N:.1f} us/pair (plain single stream)")
