"""Randomised parity of the plain HIP path (engine.flow_pair: pyramid + corner kernel + fused LK of all levels) against the
CPU oracle on small ragged sizes.  Test infrastructure (imports oracle/).   python tools/fuzz_plain_vs_oracle.py [n] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cuda_optical_flow_2_amd import engine as eng, synth
from oracle import Oracle


def run(n_cfg: int, seed: int, verbose: bool = True) -> int:
    rng = np.random.default_rng(seed)
    orc = Oracle()
    bad = 0
    for it in range(n_cfg):
        L = int(rng.integers(1, 5))
        step = 1 << (L - 1)
        w = step * int(rng.integers(1, max(2, 330 // step)))
        h = step * int(rng.integers(1, max(2, 200 // step)))
        mode = "lk_float" if rng.random() < 0.5 else "compat_cpu"
        win = int(rng.choice([3, 5, 7, 9, 13, 19, 23] + ([25] if mode == "compat_cpu" else [])))
        p, n = (synth.smooth_pair(w, h, float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), seed=it) if rng.random() < 0.7
                else synth.random_pair(w, h, seed=it))
        desc = f"{w}x{h} L{L} w{win} {mode}"
        got = eng.flow_pair(p, n, L, win, mode)
        want, _, _ = orc.flow_pair(synth.to_3ch(p), synth.to_3ch(n), L, win, mode, exact_sums=True)
        nbad = 0
        for k in range(L):
            a, b = got[k], want[k]
            same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
            nbad += int((~same).sum())
        if nbad:
            bad += 1
            print("FAIL", desc, nbad)
        elif verbose:
            print("ok  ", desc)
    return bad


if __name__ == "__main__":
    failures = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("failures:", failures)
    sys.exit(1 if failures else 0)
