#!/usr/bin/env python3
"""Time the bilateral pre-filter kernels (exact / +-1 LSB) on a 4K frame and compare the fast one with the exact one.
Usage: python tools/bil_bench.py [sigma_b ...]   (OFX_BILATERAL_LUT=0 switches the lane-table kernel off)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_optical_flow_2_amd import lib as _l

L = _l.load()
w, h = 3840, 2160
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(3)
# a smooth image plus noise (what the pre-filter sees), grey: three equal channels
yy, xx = torch.meshgrid(torch.arange(h, device="cuda"), torch.arange(w, device="cuda"), indexing="ij")
base = (128 + 60 * torch.sin(xx / 37.0) * torch.cos(yy / 23.0) + 12 * torch.randn((h, w), device="cuda", generator=g)).clamp(0, 255).to(torch.uint8)
gray = base[..., None].expand(h, w, 3).contiguous()
from cuda_optical_flow_2_amd import synth
tex = torch.from_numpy(synth.to_3ch(synth.smooth_pair(w, h, 2.0, 1.0)[1])).cuda()   # (bench.py's texture: no noise)
tex3 = (tex[..., 0].float() + 3 * torch.randn((h, w), device="cuda", generator=g)).clamp(0, 255).to(torch.uint8)[..., None].expand(h, w, 3).contiguous()
rnd = torch.randint(0, 256, (h, w), device="cuda", dtype=torch.uint8, generator=g)[..., None].expand(h, w, 3).contiguous()
colour = torch.randint(0, 256, (h, w, 3), device="cuda", dtype=torch.uint8, generator=g)


def timed(f, n):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for sb in [float(a) for a in sys.argv[1:]] or [10.0]:
    only = os.environ.get("BIL_CASES", "").split(",") if os.environ.get("BIL_CASES") else None
    for name, src, gr in (("texture", tex, tex), ("tex+n3", tex3, tex3), ("smooth", gray, gray), ("random", rnd, rnd), ("colour", colour, gray)):
        if only and name not in only: continue
        ex, fa = torch.empty_like(src), torch.empty_like(src)
        _l.check(L.ofx_bilateral_3ch(src.data_ptr(), gr.data_ptr(), ex.data_ptr(), w, h, 9, 9, 2.0, sb, st), "exact")
        _l.check(L.ofx_bilateral_3ch_fast(src.data_ptr(), gr.data_ptr(), fa.data_ptr(), w, h, 9, 9, 2.0, sb, st), "fast")
        torch.cuda.synchronize()
        d = (ex.int() - fa.int()).abs()
        t_e = timed(lambda: L.ofx_bilateral_3ch(src.data_ptr(), gr.data_ptr(), ex.data_ptr(), w, h, 9, 9, 2.0, sb, st), 5)
        t_f = timed(lambda: L.ofx_bilateral_3ch_fast(src.data_ptr(), gr.data_ptr(), fa.data_ptr(), w, h, 9, 9, 2.0, sb, st), 20)
        print(f"sigma_b {sb:5.1f} {name:8s} exact {t_e:7.1f} us  fast {t_f:7.1f} us  max|diff| {int(d.max())}  differing px {int((d > 0).sum())}", flush=True)
