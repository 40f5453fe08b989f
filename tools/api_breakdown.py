"""Where a frame of main.cu's loop goes when it runs through the drop-in gpu:: symbols: wall time of every call of the loop
(gauss_pyramid, calc_opt_flow per level), at 1080p and 4K.   python tools/api_breakdown.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuda_optical_flow_2_amd import synth
from cuda_optical_flow_2_amd.compat import GpuCompat, _p, _ptrs, _u8p
import ctypes as C

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gc = GpuCompat()
for name, (w, h, L) in {"1080p": (1920, 1080, 4), "4k": (3840, 2160, 5)}.items():
    frames = [synth.to_3ch(synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1]) for i in range(4)]
    loop = gc.frame_loop(w, h, L)
    loop.first(frames[0])
    for i in range(5):
        loop.step(frames[(i + 1) % 4])
    t = {"copyto": 0.0, "gauss_pyramid": 0.0, **{f"calc_opt_flow L{k}": 0.0 for k in range(L)}}
    t0 = time.perf_counter()
    for i in range(n):
        f = frames[(i + 2) % 4]
        a = time.perf_counter(); np.copyto(loop.cur[0], f); b = time.perf_counter(); t["copyto"] += b - a
        loop._gp(_ptrs(loop.cur, C.c_uint8), w, h, L, loop._mask, 3, 3); c = time.perf_counter(); t["gauss_pyramid"] += c - b
        fl = _ptrs(loop.flow, C.c_float)
        for k in range(L - 1, -1, -1):
            a = time.perf_counter()
            loop._cof(_p(loop.prev[k], _u8p), _p(loop.cur[k], _u8p), w >> k, h >> k, fl, k, L)
            t[f"calc_opt_flow L{k}"] += time.perf_counter() - a
        loop.prev, loop.cur = loop.cur, loop.prev
    tot = (time.perf_counter() - t0) / n * 1e3
    print(f"{name}: {tot:.3f} ms per frame; " + ", ".join(f"{k} {v / n * 1e3:.3f}" for k, v in t.items()))
    px = [(w >> k) * (h >> k) for k in range(L)]
    print(f"   bytes per frame: pyramid up {3 * px[0] / 1e6:.1f} MB + down {3 * sum(px[1:]) / 1e6:.1f} MB; flow levels: up {2 * sum(px) / 1e6:.1f} MB (channel 0 of two 3-channel images read: {6 * sum(px) / 1e6:.1f} MB), down {8 * sum(px) / 1e6:.1f} MB")
