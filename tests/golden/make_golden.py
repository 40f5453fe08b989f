"""Generate the golden fixtures in tests/golden/ from the REFERENCE's own CPU build.

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

The outputs are computed by oracle/_ref/libref_cpu.so -- OptFlowCPU.cpp, kernels.cpp and OptFlowUtils.cpp
compiled unmodified (oracle/Makefile; -ffp-contract=off; zero-filling malloc shim for determinism).
Fixtures are data only: seeded inputs and the reference's outputs.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cuda_optical_flow_2_amd import synth  # noqa: E402
from oracle import Reference  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
R = Reference()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path)} bytes")


def primitives():
    rng = np.random.default_rng(20240501)
    img = rng.integers(0, 256, (24, 40, 3), dtype=np.uint8)
    gray = R.grayscale_avg_cpu(img)
    d = {"img": img, "gray": gray}
    for nm, m in (("dx", R.Dx_3x3), ("dy", R.Dy_3x3), ("dt", R.Dt_3x3), ("gaus", R.GAUS_KERNEL_3x3)):
        d["conv1_" + nm] = R.conv_3ch_to_1ch(gray, m)
        d["conv3_" + nm] = R.conv_3ch(img, m, 3, 3)
    m5 = (rng.integers(-8, 9, 25) / 4.0).astype(np.float32)
    d["mask5"] = m5
    d["conv1_m5"] = R.conv_3ch_to_1ch(gray, m5, 5, 5)
    d["conv3_m5"] = R.conv_3ch(img, m5, 5, 5)
    d["down"] = R.downscale_gaussian(img)
    a = rng.integers(0, 256, (24, 40), dtype=np.uint8)
    b = rng.integers(0, 256, (24, 40), dtype=np.uint8)
    d["a"], d["b"] = a, b
    d["sub"] = R.sub_arr(a, b)
    for ww, wh in ((3, 3), (5, 5), (7, 7), (9, 9), (15, 15), (19, 19), (5, 9), (4, 6)):
        d[f"srm_{ww}x{wh}"] = R.srm_1ch(a, b, ww, wh)
    s = [rng.integers(-4000, 4000, (12, 20)).astype(np.int32) for _ in range(5)]
    s[0], s[1] = np.abs(s[0]), np.abs(s[1])
    for k in range(3):
        s[k][0, 0] = 0            # det == 0 -> NaN / Inf
    s[0][0, 1] = 4; s[1][0, 1] = 9; s[2][0, 1] = 6   # det == 0 with non-zero entries
    d["solve_in"] = np.stack(s)
    d["solve_f32arith"] = R.inverse_matrix(*s)
    save("primitives", **d)


def shifts():
    p, _ = synth.random_pair(48, 36, seed=7)
    img = synth.to_3ch(p)
    d = {"img": img}
    cases = [(1.3, -0.7), (-0.5, 0.5), (-3.2, 4.9), (0.0, 0.0), (float("nan"), 1.0), (1e20, 0.0), (-47.2, 0.3), (0.9999, 35.1)]
    d["uv"] = np.array(cases, np.float32)
    for i, (u, v) in enumerate(cases):
        # two coarser levels: (u,v) = 4*f2 + 2*f1 in float, coarsest first (OptFlowCPU.cpp:257-266)
        f2 = np.array([[[u / 8, v / 8]]], np.float32)
        f1 = np.array([[[u / 4, v / 4]]], np.float32)
        d[f"shift_{i}"] = R.shift_back_pyramid(img, 0, 3, [None, f1, f2])
        d[f"f1_{i}"], d[f"f2_{i}"] = f1, f2
    save("shift", **d)


def levels_and_pairs():
    d = {}
    for tag, (p, n) in (("smooth", synth.smooth_pair(64, 48, 0.6, -0.4)), ("random", synth.random_pair(64, 48, seed=3))):
        p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
        flow = [np.zeros((48, 64, 2), np.float32)]
        R.calc_optical_flow(p3, n3, flow, 0, 1)          # single level == top level: window 9 (OptFlowCPU.cpp:344)
        d[tag + "_prev"], d[tag + "_next"] = p, n
        d[tag + "_flow_single"] = flow[0]
    p, n = synth.smooth_pair(64, 48, 2.0, 1.0)
    fl, pp, npyr = R.flow_pair(synth.to_3ch(p), synth.to_3ch(n), 3)
    d["pair_prev"], d["pair_next"] = p, n
    for k in range(3):
        d[f"pair_flow_L{k}"] = fl[k]
        d[f"pair_prevpyr_L{k}"] = pp[k][:, :, 0]
        d[f"pair_nextpyr_L{k}"] = npyr[k][:, :, 0]
    save("levels", **d)


def bilateral():
    rng = np.random.default_rng(99)
    img = rng.integers(0, 256, (20, 28, 3), dtype=np.uint8)
    gray = R.grayscale_avg_cpu(img)
    save("bilateral", img=img, gray=gray,
         gk_9_2=R.generate_gaussian_kernel(2.0, 9), gk_5_1p5=R.generate_gaussian_kernel(1.5, 5),
         out_gray_9=R.bilinear_filter_3ch(gray, gray, 9, 9, 2.0, 10.0),     # main.cu:240 operating point
         out_color_5=R.bilinear_filter_3ch(img, gray, 5, 5, 1.5, 20.0))


def surface():
    """The rest of the exported surface (SURVEY 8 f4 + the cpu:: twins whose arithmetic differs from the gpu:: ones)."""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    img2 = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    d = {"img": img, "img2": img2, "mask5": R.GAUS_KERNEL_5x5}
    for ww, wh in ((3, 3), (9, 9), (5, 7), (4, 4)):
        d[f"srm3_{ww}x{wh}"] = R.srm_3ch(img, img2, ww, wh)
    # cpu::gauss_pyramid honours its mask (gpu::gauss_pyramid does not): 5x5 Gaussian, 3 levels
    pyr = R.gauss_pyramid(img, 3, R.GAUS_KERNEL_5x5, 5, 5)
    d["pyr5_L1"], d["pyr5_L2"] = pyr[1], pyr[2]
    d["down_dx"] = R.downscale_gaussian(img, R.Dx_3x3, 3, 3)          # negative / > 255 sums: float -> int -> byte
    # shift with a destination buffer that is NOT zero: pixels whose target leaves the image keep the caller's bytes
    dest0 = rng.integers(0, 256, img.shape, dtype=np.uint8)
    d["shift_dest0"] = dest0
    cases = [(2.5, -1.25), (-40.0, 3.0), (float("nan"), 0.0), (0.0, 0.0)]
    d["shift_uv"] = np.array(cases, np.float32)
    for i, (u, v) in enumerate(cases):
        f1 = np.array([[[u / 2, v / 2]]], np.float32)
        d[f"shift3_{i}"] = R.shift_back_pyramid(img, 0, 2, [None, f1], dest_init=dest0)
    g1 = rng.integers(0, 256, (10, 14), dtype=np.uint8)
    d["g1"] = g1
    d["cleanup"] = R.cleanup_outliers(g1)
    for n in (0, 1, 2):
        d[f"up1_{n}"] = R.upscale(g1, n)
        d[f"up3_{n}"] = R.upscale(img[:6, :5], n)
    save("surface", **d)


if __name__ == "__main__":
    surface()
    if "--only-surface" in sys.argv:
        sys.exit(0)
    primitives()
    shifts()
    levels_and_pairs()
    bilateral()
