"""CPU: the oracle restatement against the committed golden fixtures (outputs of the reference's own CPU build,
tests/golden/make_golden.py) and the known-answer vectors recorded in SURVEY.md section 8c."""
import numpy as np

from conftest import assert_same
from cuda_optical_flow_2_amd import synth


def test_primitives_golden(oracle, golden):
    g = golden("primitives")
    img, gray = g["img"], g["gray"]
    assert_same(oracle.grayscale_avg(img), gray, "grayscale")
    for nm, m in (("dx", oracle.Dx_3x3), ("dy", oracle.Dy_3x3), ("dt", oracle.Dt_3x3), ("gaus", oracle.GAUS_3x3)):
        assert_same(oracle.conv_3ch_to_1ch(gray, m), g["conv1_" + nm], "conv_3ch_to_1ch " + nm)
        assert_same(oracle.conv_3ch(img, m, 3, 3), g["conv3_" + nm], "conv_3ch " + nm)
    assert_same(oracle.conv_3ch_to_1ch(gray, g["mask5"], 5, 5), g["conv1_m5"], "conv 5x5")
    assert_same(oracle.conv_3ch(img, g["mask5"], 5, 5), g["conv3_m5"], "conv3 5x5")
    assert_same(oracle.downscale_gaussian(img), g["down"], "downscale")
    assert_same(oracle.sub_u8(g["a"], g["b"]), g["sub"], "sub_arr")
    for ww, wh in ((3, 3), (5, 5), (7, 7), (9, 9), (15, 15), (19, 19), (5, 9), (4, 6)):
        assert_same(oracle.srm_1ch(g["a"], g["b"], ww, wh), g[f"srm_{ww}x{wh}"], f"srm {ww}x{wh}")
    assert_same(oracle.inverse_matrix_f32arith(*g["solve_in"]), g["solve_f32arith"], "inverse_matrix")


def test_shift_golden(oracle, golden):
    g = golden("shift")
    for i in range(len(g["uv"])):
        got = oracle.shift_back_pyramid(g["img"], 0, 3, [None, g[f"f1_{i}"], g[f"f2_{i}"]])
        assert_same(got, g[f"shift_{i}"], f"shift case {i} uv={g['uv'][i]}")


def test_levels_golden(oracle, golden):
    g = golden("levels")
    for tag in ("smooth", "random"):
        p3, n3 = synth.to_3ch(g[tag + "_prev"]), synth.to_3ch(g[tag + "_next"])
        flow = [np.zeros((48, 64, 2), np.float32)]
        oracle.calc_optical_flow_cpu(p3, n3, flow, 0, 1, 9)
        assert_same(flow[0], g[tag + "_flow_single"], "single-level flow " + tag)
    fl, pp, npyr = oracle.flow_pair(synth.to_3ch(g["pair_prev"]), synth.to_3ch(g["pair_next"]), 3, 9, "compat_cpu")
    for k in range(3):
        assert_same(pp[k][:, :, 0], g[f"pair_prevpyr_L{k}"], f"prev pyramid L{k}")
        assert_same(npyr[k][:, :, 0], g[f"pair_nextpyr_L{k}"], f"next pyramid L{k}")
        assert_same(fl[k], g[f"pair_flow_L{k}"], f"3-level pipeline flow L{k}")


def test_bilateral_golden(oracle, golden):
    g = golden("bilateral")
    assert_same(oracle.generate_gaussian_kernel(2.0, 9), g["gk_9_2"], "gaussian kernel 9")
    assert_same(oracle.generate_gaussian_kernel(1.5, 5), g["gk_5_1p5"], "gaussian kernel 5")
    assert_same(oracle.bilateral_3ch(g["gray"], g["gray"], 9, 9, 2.0, 10.0), g["out_gray_9"], "bilateral grey 9x9")
    assert_same(oracle.bilateral_3ch(g["img"], g["gray"], 5, 5, 1.5, 20.0), g["out_color_5"], "bilateral colour 5x5")


def test_surface_golden(oracle, golden):
    """The rest of the exported surface (cpu::srm_3ch, cpu::gauss_pyramid with a caller's mask, the 3-channel shift into a
    non-zero destination, utils::cleanup_outliers / upscale_*): oracle restatement == the reference's own build."""
    g = golden("surface")
    img, img2 = g["img"], g["img2"]
    for ww, wh in ((3, 3), (9, 9), (5, 7), (4, 4)):
        assert_same(oracle.srm_3ch(img, img2, ww, wh), g[f"srm3_{ww}x{wh}"], f"srm_3ch {ww}x{wh}")
    pyr = oracle.gauss_pyramid(img, 3, g["mask5"], 5, 5)
    assert_same(pyr[1], g["pyr5_L1"], "5x5 pyramid L1")
    assert_same(pyr[2], g["pyr5_L2"], "5x5 pyramid L2")
    assert_same(oracle.downscale_gaussian(img, oracle.Dx_3x3, 3, 3), g["down_dx"], "downscale with a signed mask")
    for i, (u, v) in enumerate(g["shift_uv"]):
        f1 = np.array([[[u / 2, v / 2]]], np.float32)
        assert_same(oracle.shift_back_pyramid(img, 0, 2, [None, f1], dest_init=g["shift_dest0"]), g[f"shift3_{i}"], f"3ch shift {i}")
    assert_same(oracle.cleanup_outliers(g["g1"]), g["cleanup"], "cleanup_outliers")
    for n in (0, 1, 2):
        assert_same(oracle.upscale(g["g1"], n), g[f"up1_{n}"], f"upscale_1ch {n}")
        assert_same(oracle.upscale(img[:6, :5], n), g[f"up3_{n}"], f"upscale_3ch {n}")


# ---- known-answer vectors of SURVEY.md 8c(iv), verified there against the reference build -------------------------

def test_kat_gaussian_truncation(oracle):
    ones = np.ones((5, 5, 3), np.uint8)
    assert oracle.conv_3ch_to_1ch(ones, oracle.GAUS_3x3).max() == 0           # every tap truncates to 0
    c = oracle.conv_3ch_to_1ch(np.full((5, 5, 3), 200, np.uint8), oracle.GAUS_3x3)
    assert c[2, 2] == 198 and c[0, 0] == 112


def test_kat_sobel_wrap(oracle):
    row = np.array([0, 0, 100, 200, 200, 10], np.uint8)
    img = synth.to_3ch(np.tile(row, (5, 1)))
    out = oracle.conv_3ch_to_1ch(img, oracle.Dx_3x3)[2]
    # 4*(right-left) mod 256: 400 -> 144, 800 -> 32, 400 -> 144, -760 -> 8, (0-200)*4 = -800 -> 224 at the border
    want = [(4 * (int(row[min(i + 1, 5)] if i + 1 < 6 else 0) - int(row[i - 1] if i > 0 else 0))) % 256 for i in range(6)]
    assert out.tolist() == want


def test_kat_srm_clipping(oracle):
    a = np.full((5, 5), 2, np.uint8)
    b = np.full((5, 5), 3, np.uint8)
    s = oracle.srm_1ch(a, b, 3, 3)
    assert s[0, 0] == 24 and s[0, 2] == 36 and s[2, 2] == 54


def test_kat_shift(oracle):
    row = np.array([3, 6, 9, 12, 15, 18], np.uint8)
    img = synth.to_3ch(np.tile(row, (3, 1)))
    out = oracle.shift_back_pyramid(img, 0, 2, [None, np.array([[[0.5, 0.0]]], np.float32)])  # u = 2*0.5 = 1
    # last column's target is outside the image; it keeps the memcpy'd byte only when 3*pos < w*h (row 0 only)
    assert out[0, :, 0].tolist() == [6, 9, 12, 15, 18, 18]
    assert out[2, :, 0].tolist() == [6, 9, 12, 15, 18, 0]


def test_kat_solve(oracle):
    one = lambda v: np.array([[v]], np.int32)  # noqa: E731
    args = (one(50), one(40), one(10), one(-30), one(20))   # sxx, syy, sxy, sxt, syt
    good = oracle.inverse_matrix_i32(*args)[0, 0]
    assert abs(good[0] - 0.736842) < 1e-6 and abs(good[1] + 0.684211) < 1e-6
    inline = oracle.inverse_matrix_inline_cpu(*args)[0, 0]
    assert abs(inline[0] - 0.736842) < 1e-6 and abs(inline[1] + 300.526) < 1e-3   # `c` not scaled: OptFlowCPU.cpp:373-376
    z = oracle.inverse_matrix_i32(*(one(0),) * 5)[0, 0]
    assert np.isnan(z).all()


def test_float_sum_order_tolerance(oracle):
    """The exact window sum and the reference's row-major float accumulation agree to 361*2^-24 of sum|a*b|."""
    p, n = synth.random_pair(96, 64, seed=5)
    ix, iy, it, _ = oracle.level_planes(synth.to_3ch(p), synth.to_3ch(n), 19, 1, want_sums=False)
    for a, b in ((ix, ix), (ix, it), (iy, it)):
        ordered = oracle.srm_1ch_f32(a, b, 19, 19)
        exact = oracle.srm_1ch_f32(a, b, 19, 19, exact=True)
        bound = oracle.srm_1ch_f32(np.abs(a), np.abs(b), 19, 19, exact=True) * (361 * 2.0 ** -24)
        assert (np.abs(ordered.astype(np.float64) - exact.astype(np.float64)) <= bound + 1e-6).all()


def test_oracle_goldens_under_ubsan():
    """The restatement rebuilt with -fsanitize=undefined -fno-sanitize-recover=all (`make -C oracle ubsan`) runs every golden test
    of this file in a child process: one report of undefined behaviour aborts that process.  (The reference has some on this
    path -- uninitialised reads, float -> int of huge values, OptFlowCPU.cpp:268-269 -- which the restatement reaches the same
    results without.)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "ubsan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, OFX_ORACLE_SO="liboracle_ubsan.so", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.abspath(__file__), "-k", "golden and not ubsan"],
                       env=env, cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout
