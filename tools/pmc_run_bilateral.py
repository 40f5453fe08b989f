"""Tiny workload for rocprofv3 --pmc passes: the 9x9 bilateral pre-filter of main.cu:240 on a 4K grey frame (3 channels),
bit-exact and +-1 LSB kernels, a few launches each.   python tools/pmc_run_bilateral.py"""
import sys
import torch
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import lib as _l, synth
L = _l.load()
w, h = 3840, 2160
g = torch.from_numpy(synth.to_3ch(synth.smooth_pair(w, h)[0])).cuda()
out = torch.empty_like(g)
for i in range(8):
    _l.check(L.ofx_bilateral_3ch(g.data_ptr(), g.data_ptr(), out.data_ptr(), w, h, 9, 9, 2.0, 10.0, None), "bilateral")
for i in range(8):
    _l.check(L.ofx_bilateral_3ch_fast(g.data_ptr(), g.data_ptr(), out.data_ptr(), w, h, 9, 9, 2.0, 10.0, None), "bilateral_fast")
torch.cuda.synchronize()
