"""CPU: differential test of the oracle restatement against the reference's own CPU sources compiled in the build
container (oracle/_ref/libref_cpu.so).  Skipped where that build is absent."""
import numpy as np
import pytest

from conftest import assert_same
from cuda_optical_flow_2_amd import synth


@pytest.mark.parametrize("seed", [0, 1])
def test_primitives(oracle, ref, seed):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (37, 50, 3), dtype=np.uint8)
    assert_same(oracle.grayscale_avg(img), ref.grayscale_avg_cpu(img), "gray")
    g3 = oracle.grayscale_avg(img)
    for m in (oracle.Dx_3x3, oracle.Dy_3x3, oracle.GAUS_3x3, oracle.Dt_3x3):
        assert_same(oracle.conv_3ch_to_1ch(g3, m), ref.conv_3ch_to_1ch(g3, m), "conv1")
        assert_same(oracle.conv_3ch(img, m, 3, 3), ref.conv_3ch(img, m, 3, 3), "conv3")
    m5 = rng.normal(size=25).astype(np.float32)
    assert_same(oracle.conv_3ch_to_1ch(g3, m5, 5, 5), ref.conv_3ch_to_1ch(g3, m5, 5, 5), "conv 5x5 arbitrary mask")
    big = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    for a, b in zip(oracle.gauss_pyramid(big, 3), ref.gauss_pyramid(big, 3)):
        assert_same(a, b, "pyramid level")
    a = rng.integers(0, 256, (33, 41), dtype=np.uint8)
    b = rng.integers(0, 256, (33, 41), dtype=np.uint8)
    for ww, wh in ((5, 5), (7, 7), (9, 9), (15, 15), (19, 19), (3, 7), (4, 6), (1, 1), (41, 3)):
        assert_same(oracle.srm_1ch(a, b, ww, wh), ref.srm_1ch(a, b, ww, wh), f"srm {ww}x{wh}")
    assert_same(oracle.sub_u8(a, b), ref.sub_arr(a, b), "sub")


def test_shift_and_solve(oracle, ref):
    rng = np.random.default_rng(2)
    big = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    for uv in ((1.3, -0.7), (-0.5, 0.5), (np.nan, 1.0), (1e20, -1e20), (-63.5, 47.2), (0.0, 0.0)):
        fl = [None, np.array([[[uv[0], uv[1]]]], np.float32).repeat(4, 0), np.array([[[0.4, 0.9]]], np.float32)]
        for lvl in (0, 1):
            assert_same(oracle.shift_back_pyramid(big, lvl, 3, fl), ref.shift_back_pyramid(big, lvl, 3, fl), f"shift {uv} L{lvl}")
    s = [rng.integers(-5000, 5000, (20, 30)).astype(np.int32) for _ in range(5)]
    s[0], s[1] = np.abs(s[0]), np.abs(s[1])
    for k in range(3):
        s[k][0, 0] = 0
    assert_same(oracle.inverse_matrix_f32arith(*s), ref.inverse_matrix(*s), "inverse_matrix")


@pytest.mark.parametrize("gen", ["smooth", "random"])
@pytest.mark.parametrize("shape", [(64, 48, 3), (160, 120, 3), (96, 64, 2)])
def test_full_pairs(oracle, ref, gen, shape):
    w, h, levels = shape
    p, n = synth.smooth_pair(w, h) if gen == "smooth" else synth.random_pair(w, h)
    p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
    fo, _, _ = oracle.flow_pair(p3, n3, levels, 9, "compat_cpu")
    fr, _, _ = ref.flow_pair(p3, n3, levels)
    for k in range(levels):
        assert_same(fo[k], fr[k], f"flow L{k}")


def test_bilateral(oracle, ref):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (20, 24, 3), dtype=np.uint8)
    gs = oracle.grayscale_avg(img)
    assert_same(oracle.generate_gaussian_kernel(2.0, 9), ref.generate_gaussian_kernel(2.0, 9), "gaussian kernel")
    assert_same(oracle.bilateral_3ch(gs, gs, 9, 9, 2, 10), ref.bilinear_filter_3ch(gs, gs, 9, 9, 2, 10), "bilateral grey")
    assert_same(oracle.bilateral_3ch(img, gs, 5, 5, 1.5, 20), ref.bilinear_filter_3ch(img, gs, 5, 5, 1.5, 20), "bilateral colour")


def test_leftover_surface(oracle, ref):
    """cpu::srm_3ch (its `>` bounds test), cpu::gauss_pyramid with arbitrary masks, the 3-channel shift into a dirty
    destination, utils::cleanup_outliers / upscale_*."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (21, 34, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (21, 34, 3), dtype=np.uint8)
    for ww, wh in ((3, 3), (9, 9), (7, 5), (2, 6), (1, 1), (35, 3)):
        assert_same(oracle.srm_3ch(a, b, ww, wh), ref.srm_3ch(a, b, ww, wh), f"srm_3ch {ww}x{wh}")
    big = rng.integers(0, 256, (48, 64, 3), dtype=np.uint8)
    for mask, mw, mh in ((ref.GAUS_KERNEL_5x5, 5, 5), (rng.normal(size=9).astype(np.float32), 3, 3), (np.full(12, 0.1, np.float32), 4, 3)):
        for x, y in zip(oracle.gauss_pyramid(big, 3, mask, mw, mh), ref.gauss_pyramid(big, 3, mask, mw, mh)):
            assert_same(x, y, f"pyramid with a {mw}x{mh} mask")
    dirty = rng.integers(0, 256, big.shape, dtype=np.uint8)
    for uv in ((2.5, -1.25), (-70.0, 3.0), (np.nan, 0.0), (0.0, 0.0), (0.0, 47.9)):
        fl = [None, np.array([[[uv[0] / 2, uv[1] / 2]]], np.float32)]
        assert_same(oracle.shift_back_pyramid(big, 0, 2, fl, dest_init=dirty), ref.shift_back_pyramid(big, 0, 2, fl, dest_init=dirty), f"shift {uv}")
    g1 = rng.integers(0, 256, (9, 13), dtype=np.uint8)
    assert_same(oracle.cleanup_outliers(g1), ref.cleanup_outliers(g1), "cleanup_outliers")
    for n in (0, 1, 3):
        assert_same(oracle.upscale(g1, n), ref.upscale(g1, n), f"upscale_1ch {n}")
        assert_same(oracle.upscale(a[:5, :7], n), ref.upscale(a[:5, :7], n), f"upscale_3ch {n}")
