"""PCIe-inclusive rates: the same pair with host buffers crossing the boundary."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import engine, synth
from cuda_optical_flow_2_amd.compat import GpuCompat

for (nm, w, h, L, win) in (("1080p", 1920, 1080, 4, 7), ("4k", 3840, 2160, 5, 9)):
    p, n = synth.smooth_pair(w, h)
    s = engine.Session(w, h, L, win, "lk_float")
    s.push_frame_host(p)
    out = None
    def pair_host(download):
        s.set_frame_host(n); s.build_pyramid(); s.run_flow()
        if download:
            return [s.flow_host(k) for k in range(L)]
        torch.cuda.synchronize()
    for download in (False, True):
        pair_host(download)
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps): pair_host(download)
        dt = (time.perf_counter() - t0) / reps
        print(f"{nm} session, frame uploaded from pageable host memory{', all flow levels downloaded' if download else ''}: "
              f"{dt*1e3:.2f} ms/pair = {w*h/dt/1e6:.0f} Mpix/s")
    s.close()
    if nm == "1080p":
        g = GpuCompat()
        p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
        g.flow_pair(p3, n3, L)
        t0 = time.perf_counter()
        g.flow_pair(p3, n3, L)
        dt = time.perf_counter() - t0
        print(f"{nm} gpu:: drop-in surface (host pointers per call, window 19, both pyramids): {dt*1e3:.1f} ms/pair = {w*h/dt/1e6:.0f} Mpix/s")
