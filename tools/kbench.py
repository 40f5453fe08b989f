"""Kernel micro-benchmark: per-level time of shift+LK through the session, and whole-pair time."""
import sys, os
import numpy as np, torch
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import engine, synth

cfgs = {"4k": (3840, 2160, 5, 9), "1080p": (1920, 1080, 4, 7), "8k": (7680, 4320, 6, 15)}
names = sys.argv[1:] or ["4k", "1080p"]
for nm in names:
    w, h, L, win = cfgs[nm]
    p, n = synth.smooth_pair(w, h)
    s = engine.Session(w, h, L, win, "lk_float")
    s.push_frame_host(p)
    s.set_frame_host(n); s.build_pyramid(); s.run_flow(); torch.cuda.synchronize()
    tn = torch.from_numpy(n).cuda()
    print("   shift vectors (u,v) per level:", [tuple(round(float(x), 3) for x in s.uv(k).cpu().tolist()) for k in range(L - 1)])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def timeit(fn, reps=30):
        best = 1e9
        for _ in range(3):
            e0.record()
            for i in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / reps * 1e3)
        return best
    def pair():
        s.set_frame_device(tn); s.build_pyramid(); s.run_flow()
    t = timeit(pair)
    print(f"{nm}: pair {t:.1f} us  {w*h/t:.0f} Mpix/s", end=" | ")
    print(f"pyr {timeit(lambda: s.build_pyramid()):.1f} corner {timeit(lambda: s.corner_flows()):.1f} levels {timeit(lambda: s.run_levels()):.1f}", end=" | ")
    for k in range(L):
        tk = timeit(lambda: s.run_level(k))
        print(f"L{k} {tk:.1f}", end=" ")
    s.timing(64)
    for i in range(64): s.run_level(0)
    torch.cuda.synchronize()
    avg, mn, cnt = s.timing_read()
    print(f"| L0 lk kernel avg {avg:.1f} min {mn:.1f} us -> {w*h*10/avg/1e3:.0f} GB/s ({w*h*10/avg/1e3/80:.1f}% of 8 TB/s)", end=" ")
    s.timing(64)
    for i in range(64): s.run_levels()
    torch.cuda.synchronize()
    avg, mn, cnt = s.timing_read()
    tot = sum((w >> k) * (h >> k) for k in range(L))
    print(f"| all-level lk launch avg {avg:.1f} min {mn:.1f} us -> {tot*10/avg/1e3:.0f} GB/s ({tot*10/avg/1e3/80:.1f}%)")
    s.close()
