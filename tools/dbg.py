import sys
import numpy as np
sys.path.insert(0, ".")
from cuda_optical_flow_2_amd import engine, synth
from oracle import Oracle
O = Oracle()
np.set_printoptions(linewidth=250)
for px in (5, 3, 4):
    p = np.zeros((7, 16), np.uint8); p[3, px] = 100
    n = p.copy()
    ix, iy, it, sums = O.level_planes(synth.to_3ch(p), synth.to_3ch(n), 3, 0, exact_sums=True)
    got = engine.lk_level(p, n, 3, "compat_cpu", want_sums=True)
    print("pixel at x=", px); print("oracle Ix\n", ix.astype(int)); print("want Sxx\n", sums[0].astype(int)); print("got Sxx\n", got[0])
