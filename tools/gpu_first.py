"""First-contact GPU check: parity of the fused level against the oracle on small inputs, then a 4K timing."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import engine, synth, lib
from oracle import Oracle

O = Oracle()
print("device", torch.cuda.get_device_name(0), "ofx devices", lib.load().ofx_device_count())

def cmp(name, a, b):
    a = np.asarray(a); b = np.asarray(b)
    if a.dtype.kind == "f":
        same = np.array_equal(a, b, equal_nan=True)
        nbad = int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum())
    else:
        same = np.array_equal(a, b); nbad = int((a != b).sum())
    print(("OK   " if same else "FAIL ") + name, "" if same else f"mismatches={nbad}/{a.size}")
    return same

for (w, h) in ((64, 48), (300, 37), (517, 64)):
    for gen in ("smooth", "random"):
        p, n = (synth.smooth_pair(w, h, 0.6, -0.4) if gen == "smooth" else synth.random_pair(w, h))
        p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
        for win in (3, 5, 7, 9, 15, 19):
            for mode, m in (("compat_cpu", 0), ("lk_float", 1)):
                _, _, _, sums = O.level_planes(p3, n3, win, m, exact_sums=True)
                got = engine.lk_level(p, n, win, mode, want_sums=True)
                ok = cmp(f"sums {w}x{h} {gen} w{win} {mode}", got, sums.astype(np.int64).astype(np.int32))
                fl = [np.zeros((h, w, 2), np.float32)]
                if m == 0:
                    O.calc_optical_flow_cpu(p3, n3, fl, 0, 1, win)
                else:
                    O.calc_opt_flow_gpu(p3, n3, fl, 0, 1, win, exact_sums=True)
                gotf = engine.lk_level(p, n, win, mode)
                cmp(f"flow {w}x{h} {gen} w{win} {mode}", gotf, fl[0])
                if not ok:
                    d = np.argwhere(got != sums.astype(np.int64).astype(np.int32))
                    print("   first bad (plane,y,x):", d[:5].tolist())

# pyramid / shift
big = synth.random_pair(128, 96)[0]
cmp("downsample", engine.downsample_1ch(big), O.downscale_gaussian(synth.to_3ch(big))[:, :, 0])
for uv in ((1.3, -0.7), (-0.5, 0.5), (float("nan"), 1.0), (1e20, 0.0), (-3.2, 4.9), (0.0, 0.0)):
    fl = [None, np.array([[[uv[0], uv[1]]]], np.float32)]
    cmp(f"shift {uv}", engine.shift_1ch(big, uv), O.shift_back_pyramid(synth.to_3ch(big), 0, 2, fl)[:, :, 0])

# whole pair
for mode in ("compat_cpu", "lk_float"):
    p, n = synth.smooth_pair(256, 192)
    got = engine.flow_pair(p, n, 3, 9, mode)
    ref, _, _ = O.flow_pair(synth.to_3ch(p), synth.to_3ch(n), 3, 9, mode, exact_sums=True)
    for k in range(3):
        cmp(f"pair {mode} L{k}", got[k], ref[k])

# timing
for (w, h, L, win) in ((1920, 1080, 4, 7), (3840, 2160, 5, 9)):
    p, n = synth.smooth_pair(w, h)
    s = engine.Session(w, h, L, win, "lk_float")
    s.push_frame_host(p)
    s.set_frame_host(n); s.build_pyramid(); s.run_flow(); torch.cuda.synchronize()
    tn = torch.from_numpy(n).cuda()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        e0.record()
        for i in range(20):
            s.set_frame_device(tn); s.build_pyramid(); s.run_flow()
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{w}x{h} L{L} w{win}: {ms*1e3:.1f} us/pair  {w*h/ms/1e3:.0f} Mpix/s")
    # level-0 kernel alone
    for rep in range(2):
        e0.record()
        for i in range(20):
            s.run_level(0)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"   level0 shift+lk: {ms*1e3:.1f} us  -> {w*h*10/ms/1e6:.0f} GB/s algorithmic (10 B/px)")
    s.close()
