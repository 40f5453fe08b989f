"""Timeline of ONE stream-kernel launch, from per-wave start/end timestamps the kernel records (ofx_debug_stream_trace):
when do the LK waves start and end, when do the pyramid blocks run.   python tools/stream_timeline.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_optical_flow_2_amd import engine, lib, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h, L, win = 3840, 2160, 5, 9
frames = [torch.from_numpy(synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1]).cuda() for i in range(4)]
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
s = engine.Session(w, h, L, win, "lk_float", stream_batch=B)
s.stream_begin()
for i in range(10 * B):
    s.stream_submit(frames[i % 4])
torch.cuda.synchronize()
cap = 16384
buf = torch.zeros(8 * cap, dtype=torch.int64, device="cuda")
Lib = lib.load()
Lib.ofx_debug_stream_trace(buf.data_ptr(), cap, None)
for i in range(B):
    s.stream_submit(frames[i % 4])
torch.cuda.synchronize()
first = (C.c_int * 9)()
Lib.ofx_debug_stream_trace(None, 0, first)
first = list(first)
t = buf.cpu().numpy().reshape(-1, 2).astype(np.float64)
nb = first[-1]
t = t[: 4 * nb]
ok = t[:, 1] > 0
t0 = t[ok, 0].min()
us = (t - t0) / 100.0  # 100 MHz
def rng(a, b):
    m = ok[4 * a: 4 * b]
    x = us[4 * a: 4 * b][m]
    return x
lk = rng(4, first[0])
print(f"blocks: corner 0..3, LK 4..{first[0]}, pyramid stages {first}")
print(f"LK waves {len(lk)}: start min/median/max {lk[:,0].min():.1f}/{np.median(lk[:,0]):.1f}/{lk[:,0].max():.1f} us, "
      f"end min/median/max {lk[:,1].min():.1f}/{np.median(lk[:,1]):.1f}/{lk[:,1].max():.1f} us, duration median {np.median(lk[:,1]-lk[:,0]):.1f}")
for i in range(8):
    if first[i + 1] > first[i]:
        p = rng(first[i], first[i + 1])
        print(f"pyramid stage {i}: {first[i+1]-first[i]} blocks, start min/median/max {p[:,0].min():.1f}/{np.median(p[:,0]):.1f}/{p[:,0].max():.1f}, "
              f"end max {p[:,1].max():.1f}, block duration median {np.median(p[:,1]-p[:,0]):.2f} p90 {np.percentile(p[:,1]-p[:,0],90):.2f} us")
c = rng(0, 4)
print(f"corner waves: {[(round(a,1), round(b,1)) for a, b in c if b > a]}")
print(f"kernel span {us[ok,1].max():.1f} us")
hist, edges = np.histogram(lk[:, 1], bins=12)
print("LK end-time histogram:", [(round(e, 0), int(n)) for e, n in zip(edges[:-1], hist)])
