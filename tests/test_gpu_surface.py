"""GPU: the whole exported C++ surface through its mangled names -- namespace cpu (include/OptFlowCpu.hpp), namespace utils
and the gpu:: leftovers -- against the goldens generated from the reference's own CPU build (tests/golden/) and the oracle.
The cpu:: functions are the reference's CPU call surface executed on the device (csrc/compat_cpu.cpp), so every one of
them is held to the reference's bits."""
import numpy as np
import pytest

from conftest import assert_same
from cuda_optical_flow_2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def cpu():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cuda_optical_flow_2_amd.compat import CpuCompat

    return CpuCompat()


@pytest.fixture(scope="module")
def gpu():
    from cuda_optical_flow_2_amd.compat import GpuCompat

    return GpuCompat()


@pytest.fixture(scope="module")
def utils():
    from cuda_optical_flow_2_amd.compat import UtilsCompat

    return UtilsCompat()


def test_cpu_namespace_primitives_golden(cpu, gpu, golden):
    g = golden("primitives")
    img, gray = g["img"], g["gray"]
    assert_same(cpu.grayscale_avg_cpu(img), gray, "cpu::grayscale_avg_cpu")
    for nm, m in (("dx", gpu.Dx_3x3), ("dy", gpu.Dy_3x3), ("dt", gpu.Dt_3x3), ("gaus", gpu.GAUS_KERNEL_3x3)):
        assert_same(cpu.conv_3ch_to_1ch(gray, m), g["conv1_" + nm], f"cpu::conv_3ch_to_1ch {nm}")
        assert_same(cpu.conv_3ch(img, m, 3, 3), g["conv3_" + nm], f"cpu::conv_3ch {nm}")
    assert_same(cpu.conv_3ch_to_1ch(gray, g["mask5"], 5, 5), g["conv1_m5"], "cpu::conv_3ch_to_1ch 5x5")
    assert_same(cpu.conv_3ch(img, g["mask5"], 5, 5), g["conv3_m5"], "cpu::conv_3ch 5x5")
    assert_same(cpu.downscale_gaussian(img, gpu.GAUS_KERNEL_3x3), g["down"], "cpu::downscale_gaussian")
    assert_same(cpu.sub_arr(g["a"], g["b"]), g["sub"], "cpu::sub_arr (main.cu:64)")
    for ww, wh in ((3, 3), (5, 5), (7, 7), (9, 9), (15, 15), (19, 19), (5, 9), (4, 6)):
        assert_same(cpu.srm_1ch(g["a"], g["b"], ww, wh), g[f"srm_{ww}x{wh}"], f"cpu::srm_1ch {ww}x{wh}")
    assert_same(cpu.inverse_matrix(*g["solve_in"]), g["solve_f32arith"], "cpu::inverse_matrix (float arithmetic)")


def test_cpu_namespace_surface_golden(cpu, golden):
    g = golden("surface")
    img, img2 = g["img"], g["img2"]
    for ww, wh in ((3, 3), (9, 9), (5, 7), (4, 4)):
        assert_same(cpu.srm_3ch(img, img2, ww, wh), g[f"srm3_{ww}x{wh}"], f"cpu::srm_3ch {ww}x{wh}")
    pyr = cpu.gauss_pyramid(img, 3, g["mask5"], 5, 5)   # cpu::gauss_pyramid honours its mask
    assert_same(pyr[1], g["pyr5_L1"], "cpu::gauss_pyramid 5x5 L1")
    assert_same(pyr[2], g["pyr5_L2"], "cpu::gauss_pyramid 5x5 L2")
    dx = np.array([-1, 0, 1, -2, 0, 2, -1, 0, 1], np.float32)
    assert_same(cpu.downscale_gaussian(img, dx), g["down_dx"], "cpu::downscale_gaussian with a signed mask")
    for i, (u, v) in enumerate(g["shift_uv"]):
        f1 = np.array([[[u / 2, v / 2]]], np.float32)
        assert_same(cpu.shift_back_pyramid(img, 0, 2, [None, f1], dest_init=g["shift_dest0"]), g[f"shift3_{i}"],
                    f"cpu::shift_back_pyramid case {i} into a dirty destination")


def test_cpu_shift_golden(cpu, golden):
    g = golden("shift")
    for i in range(len(g["uv"])):
        got = cpu.shift_back_pyramid(g["img"], 0, 3, [None, g[f"f1_{i}"], g[f"f2_{i}"]])
        assert_same(got, g[f"shift_{i}"], f"cpu::shift_back_pyramid case {i} uv={g['uv'][i]}")


def test_cpu_calc_optical_flow_golden(cpu, gpu, golden):
    """cpu::calc_optical_flow == the reference's own CPU output: single levels and the 3-level pipeline of main.cu:256-262
    with the cpu:: alternates swapped in (main.cu:251,261)."""
    g = golden("levels")
    for tag in ("smooth", "random"):
        p3, n3 = synth.to_3ch(g[tag + "_prev"]), synth.to_3ch(g[tag + "_next"])
        flow = [np.zeros((48, 64, 2), np.float32)]
        cpu.calc_optical_flow(p3, n3, flow, 0, 1)
        assert_same(flow[0], g[tag + "_flow_single"], "cpu::calc_optical_flow single level " + tag)
    fl, pp, npyr = cpu.flow_pair(synth.to_3ch(g["pair_prev"]), synth.to_3ch(g["pair_next"]), 3, gpu.GAUS_KERNEL_3x3)
    for k in range(3):
        assert_same(pp[k][:, :, 0], g[f"pair_prevpyr_L{k}"], f"prev pyramid L{k}")
        assert_same(npyr[k][:, :, 0], g[f"pair_nextpyr_L{k}"], f"next pyramid L{k}")
        assert_same(fl[k], g[f"pair_flow_L{k}"], f"cpu:: 3-level pipeline flow L{k}")


def test_cpu_bilinear_filter_golden(cpu, golden):
    g = golden("bilateral")
    assert_same(cpu.bilinear_filter_3ch(g["gray"], g["gray"], 9, 9, 2.0, 10.0), g["out_gray_9"], "cpu::bilinear_filter_3ch 9x9 (main.cu:239)")
    assert_same(cpu.bilinear_filter_3ch(g["img"], g["gray"], 5, 5, 1.5, 20.0), g["out_color_5"], "cpu::bilinear_filter_3ch 5x5")


def test_cpu_namespace_larger_inputs_vs_oracle(cpu, oracle):
    """sizes that span several blocks and odd extents, against the oracle (itself pinned to the reference build)"""
    rng = np.random.default_rng(23)
    a = rng.integers(0, 256, (67, 301, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (67, 301, 3), dtype=np.uint8)
    assert_same(cpu.srm_3ch(a, b, 9, 9), oracle.srm_3ch(a, b, 9, 9), "cpu::srm_3ch 301x67")
    assert_same(cpu.srm_3ch(a, b, 2, 6), oracle.srm_3ch(a, b, 2, 6), "cpu::srm_3ch even window")
    big = rng.integers(0, 256, (128, 520, 3), dtype=np.uint8)
    m = rng.normal(size=25).astype(np.float32)
    for x, y in zip(cpu.gauss_pyramid(big, 4, m, 5, 5), oracle.gauss_pyramid(big, 4, m, 5, 5)):
        assert_same(x, y, "cpu::gauss_pyramid arbitrary 5x5 mask")
    dirty = rng.integers(0, 256, big.shape, dtype=np.uint8)
    for uv in ((2.5, -1.25), (-519.0, 3.0), (float("nan"), 0.0), (1e20, 0.0), (0.0, 127.5), (-0.999, -0.999)):
        fl = [None, np.array([[[uv[0] / 2, uv[1] / 2]]], np.float32)]
        assert_same(cpu.shift_back_pyramid(big, 0, 2, fl, dest_init=dirty), oracle.shift_back_pyramid(big, 0, 2, fl, dest_init=dirty), f"shift {uv}")
    x = rng.integers(0, 256, (40, 1000), dtype=np.uint8)
    y = rng.integers(0, 256, (40, 1000), dtype=np.uint8)
    assert_same(cpu.sub_arr(x, y), oracle.sub_u8(x, y), "cpu::sub_arr")


def test_utils_namespace_golden(utils, golden, oracle):
    g = golden("surface")
    assert_same(utils.cleanup_outliers(g["g1"]), g["cleanup"], "utils::cleanup_outliers")
    for n in (0, 1, 2):
        assert_same(utils.upscale(g["g1"], n), g[f"up1_{n}"], f"utils::upscale_1ch n={n}")
        assert_same(utils.upscale(g["img"][:6, :5], n), g[f"up3_{n}"], f"utils::upscale_3ch n={n}")
    b = golden("bilateral")
    assert_same(utils.generate_gaussian_kernel(2.0, 9), b["gk_9_2"], "utils::generate_gaussian_kernel")
    assert_same(utils.generate_gaussian_kernel(1.5, 5), b["gk_5_1p5"], "utils::generate_gaussian_kernel")


def test_gpu_conv_1d_3ch_matches_restatement(gpu, oracle):
    """gpu::conv_1d_3ch (the reference's practice kernel, OptFlowGpu.cu:1134-1189) vs its restatement; taps past the end of
    the buffer -- which the reference reads -- are skipped by both (documented deviation)."""
    rng = np.random.default_rng(4)
    for (h, w) in ((7, 9), (33, 130), (1, 5)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert_same(gpu.conv_1d_3ch(img), oracle.conv_1d_3ch(img), f"gpu::conv_1d_3ch {w}x{h}")


def test_wrappers_do_not_allocate_in_the_steady_state(gpu):
    """The host-pointer wrappers carve their device buffers out of a cached per-thread arena (compat_scratch.h): after the
    first call of a given size, repeating it must leave the device's free memory unchanged."""
    import torch

    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (480, 640, 3), dtype=np.uint8)
    gpu.grayscale_avg(img)
    gpu.gauss_pyramid(img, 4)
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(3):
        gpu.grayscale_avg(img)
        gpu.gauss_pyramid(img, 4)
        assert torch.cuda.mem_get_info()[0] == free0


def test_replay_of_main_cu_frame_loop(oracle, tmp_path):
    """examples/replay_main.cpp -- main.cu:192-272's call sequence through the gpu:: symbols, linked against libofx_hip.so
    by __graft_entry__.build() -- run as a program on three raw frames: the composed level-0 field it writes (main.cu:138-147
    via ofx_compose_flow_host) must equal the oracle's restatement of that loop: grayscale -> bilateral 9x9 (2, 10) ->
    pyramid -> calc_opt_flow (window 19, Dt_3x3) per level -> composition."""
    import os
    import subprocess

    from cuda_optical_flow_2_amd import build as hip_build, lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "replay_main")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(exe + ".cpp"):
        subprocess.check_call([hip_build.hipcc(), "-std=c++17", "-I" + os.path.join(root, "include"), exe + ".cpp", "-L" + lib.PKG, "-lofx_hip",
                               "-Wl,-rpath," + lib.PKG, "-o", exe])
    w, h, levels, nf = 320, 240, 4, 3
    rng = np.random.default_rng(8)
    frames = []
    for i in range(nf + 1):
        g = synth.smooth_pair(w, h, 1.5 * i, -0.75 * i, seed=77)[1].astype(np.int32)
        # a colour image whose channel mean is not any single channel (exercises grayscale_avg)
        f = np.stack([np.clip(g + 9, 0, 255), np.clip(g - 7, 0, 255), g], axis=2).astype(np.uint8)
        frames.append(f)
    raw, fld = tmp_path / "frames.raw", tmp_path / "field.raw"
    raw.write_bytes(b"".join(f.tobytes() for f in frames))
    r = subprocess.run([exe, str(w), str(h), str(nf), str(raw), str(fld)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    got = np.fromfile(fld, np.float32).reshape(nf, h, w, 2)

    prev_pyr = oracle.gauss_pyramid(oracle.grayscale_avg(frames[0]), levels)          # main.cu:198-209: the first frame is not filtered
    for f in range(1, nf + 1):
        gray = oracle.grayscale_avg(frames[f])
        pyr = oracle.gauss_pyramid(oracle.bilateral_3ch(gray, gray, 9, 9, 2.0, 10.0), levels)
        flow = [np.zeros((h >> k, w >> k, 2), np.float32) for k in range(levels)]
        for k in range(levels - 1, -1, -1):
            oracle.calc_opt_flow_gpu(prev_pyr[k], pyr[k], flow, k, levels, 19, exact_sums=True)
        assert_same(got[f - 1], oracle.compose_flow(flow, levels, 0), f"replay frame {f}: composed level-0 field")
        prev_pyr = pyr
    assert r.stdout.count("fnv") == nf


@pytest.mark.parametrize("size", [(203, 77), (64, 4), (5, 3), (130, 131)])
def test_fast_bilateral_filter_is_within_one_lsb(gpu, cpu, oracle, golden, size):
    """ofx_bilateral_3ch_fast (float accumulators, v_exp_f32 range weights, quotient around the centre value) against the
    oracle's bit-exact filter: every byte within +-1 (SURVEY 8c's tolerance for this stage), on the sizes and windows of the
    exact kernel's test, colour and grey, through the gpu:: / cpu:: wrappers with the process-wide switch on -- and the switch
    off again gives the exact bytes."""
    from cuda_optical_flow_2_amd import lib

    L = lib.load()
    w, h = size
    rng = np.random.default_rng(w * 1000 + h + 7)
    colour = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    grey = oracle.grayscale_avg(colour)
    smooth = synth.to_3ch(synth.smooth_pair(w, h, 0, 0, seed=5)[1])   # (natural-ish content: the quotient sits near the centre value)
    assert L.ofx_bilateral_wrappers_fast(1) == 0
    try:
        worst, differ = 0, 0
        # (the last three: a range Gaussian so wide that a sentinel grey value of -4096 for the out-of-image taps no longer gave
        # them a zero weight -- ADVICE r03: border pixels were pulled by whole grey levels; 5e4 falls back to the exact kernel)
        for (ww, wh, ss, sb) in ((9, 9, 2.0, 10.0), (5, 5, 1.5, 20.0), (7, 3, 1.0, 5.0), (13, 13, 3.0, 40.0), (9, 9, 2.0, 400.0), (5, 5, 1.0, 3000.0),
                                 (5, 5, 1.0, 5.0e4)):
            for src, gr, what in ((grey, grey, "grey"), (colour, grey, "colour"), (smooth, smooth, "smooth grey")):
                got = gpu.bilinear_filter(src, gr, ww, wh, ss, sb).astype(np.int32)
                want = oracle.bilateral_3ch(src, gr, ww, wh, ss, sb).astype(np.int32)
                d = np.abs(got - want)
                assert d.max() <= 1, f"{what} {ww}x{wh} at {w}x{h}: off by {d.max()}"
                worst, differ = max(worst, int(d.max())), differ + int((d != 0).sum())
        got = cpu.bilinear_filter_3ch(colour, grey, 9, 9, 2.0, 10.0).astype(np.int32)
        assert np.abs(got - oracle.bilateral_3ch(colour, grey, 9, 9, 2.0, 10.0).astype(np.int32)).max() <= 1
    finally:
        assert L.ofx_bilateral_wrappers_fast(0) == 1
    assert_same(gpu.bilinear_filter(grey, grey, 9, 9, 2.0, 10.0), oracle.bilateral_3ch(grey, grey, 9, 9, 2.0, 10.0), "exact again")


@pytest.mark.parametrize("size", [(203, 77), (64, 4), (5, 3), (3, 1), (130, 131), (261, 35)])
def test_fast_bilateral_filter_with_the_range_table_is_within_one_lsb(oracle, size):
    """The grey image as its own source (one device pointer for src and gray, main.cu:240) takes bilateral_lut_kernel: range weights
    out of an LDS table indexed by |d|, a 2 x 2 pixel block per lane, the tile loaded three dwords = four pixels at a time.  Every
    odd square window up to 13, sizes that are not multiples of the 128 x 16 tile or of four pixels (the byte path at the image's
    first and last bytes), a wide range Gaussian (no clamp at |d| = 63 or anywhere), and an image whose channels differ under
    that one pointer (the tile-wise check sends it down the exponential path): within +-1 of the oracle's bit-exact filter."""
    import torch

    from cuda_optical_flow_2_amd import lib

    L = lib.load()
    w, h = size
    rng = np.random.default_rng(w * 77 + h)
    noise = oracle.grayscale_avg(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    smooth = synth.to_3ch(synth.smooth_pair(max(w, 8), max(h, 8), 0, 0, seed=9)[1])[:h, :w].copy()
    colour = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    half = smooth.copy()
    half[:, w // 2:, 1] ^= 0x10  # grey tiles and tiles with colour in one image
    st = torch.cuda.current_stream().cuda_stream
    for (ww, ss, sb) in ((3, 0.8, 10.0), (5, 1.5, 20.0), (7, 1.0, 5.0), (9, 2.0, 10.0), (11, 2.5, 3.0), (13, 3.0, 40.0), (9, 2.0, 400.0), (9, 2.0, 1.5)):
        for img, what in ((noise, "noise"), (smooth, "smooth"), (colour, "colour"), (half, "half colour")):
            d_img = torch.from_numpy(img).cuda()
            d_out = torch.full_like(d_img, 0x5a)
            lib.check(L.ofx_bilateral_3ch_fast(d_img.data_ptr(), d_img.data_ptr(), d_out.data_ptr(), w, h, ww, ww, ss, sb, st), "fast")
            got = d_out.cpu().numpy().astype(np.int32)
            want = oracle.bilateral_3ch(img, img, ww, ww, ss, sb).astype(np.int32)
            d = np.abs(got - want)
            assert d.max() <= 1, f"{what} {ww}x{ww} sigma_b {sb} at {w}x{h}: off by {d.max()} at {np.argwhere(d > 1)[:4].tolist()}"


@pytest.mark.parametrize("size", [(203, 77), (64, 4), (5, 3), (3, 1), (130, 131), (261, 35)])
def test_bilateral_filter_of_an_image_that_is_its_own_grey_image_is_bit_exact(oracle, size):
    """ofx_bilateral_3ch with one device pointer for src and gray and a square window (main.cu:240) takes bilateral_exact_own_kernel
    (one int per pixel in the tile, a 2 x 2 block per lane, the reference's double operations in the reference's order): the
    oracle's bytes for every odd window up to 13, sizes off the 128 x 32 tile and off four pixels, narrow and wide range Gaussians
    (denormal table entries included: sigma_b 1.5), images whose channels are equal, differ everywhere, or differ in one half."""
    import torch

    from cuda_optical_flow_2_amd import lib

    L = lib.load()
    w, h = size
    rng = np.random.default_rng(w * 79 + h)
    noise = oracle.grayscale_avg(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    smooth = synth.to_3ch(synth.smooth_pair(max(w, 8), max(h, 8), 0, 0, seed=9)[1])[:h, :w].copy()
    colour = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    half = smooth.copy()
    half[:, w // 2:, 1] ^= 0x10
    st = torch.cuda.current_stream().cuda_stream
    for (ww, ss, sb) in ((3, 0.8, 10.0), (5, 1.5, 20.0), (7, 1.0, 5.0), (9, 2.0, 10.0), (11, 2.5, 3.0), (13, 3.0, 40.0), (9, 2.0, 400.0), (9, 2.0, 1.5)):
        for img, what in ((noise, "noise"), (smooth, "smooth"), (colour, "colour"), (half, "half colour")):
            d_img = torch.from_numpy(img).cuda()
            d_out = torch.full_like(d_img, 0x5a)
            lib.check(L.ofx_bilateral_3ch(d_img.data_ptr(), d_img.data_ptr(), d_out.data_ptr(), w, h, ww, ww, ss, sb, st), "exact")
            assert_same(d_out.cpu().numpy(), oracle.bilateral_3ch(img, img, ww, ww, ss, sb), f"{what} {ww}x{ww} sigma_b {sb} at {w}x{h}")


def test_grayscale_sixteen_pixels_per_thread_equals_the_oracle(oracle):
    """ofx_grayscale_avg_3ch takes three 16-byte pieces = sixteen pixels per thread where both images are 16-byte aligned, the
    byte-wise kernel for the last n % 16 pixels and for unaligned images: sizes around the multiples of 16, an image that starts
    3 bytes into its allocation, every byte against the oracle (OptFlowCPU.cpp:27-28)."""
    import torch

    from cuda_optical_flow_2_amd import lib

    L = lib.load()
    st = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(21)
    for (w, h) in ((203, 77), (64, 4), (5, 3), (16, 1), (17, 1), (1, 1), (1920, 3), (255, 255)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        want = oracle.grayscale_avg(img)
        for shift in (0, 3):
            buf = torch.zeros(img.size + 64, dtype=torch.uint8, device="cuda")
            out = torch.full((img.size + 64,), 0x5a, dtype=torch.uint8, device="cuda")
            buf[shift:shift + img.size] = torch.from_numpy(img.reshape(-1)).cuda()
            lib.check(L.ofx_grayscale_avg_3ch(buf.data_ptr() + shift, out.data_ptr() + shift, w, h, st), "grayscale")
            got = out.cpu().numpy()
            assert_same(got[shift:shift + img.size].reshape(h, w, 3), want, f"{w}x{h} at byte offset {shift}")
            assert (got[:shift] == 0x5a).all() and (got[shift + img.size:] == 0x5a).all(), f"{w}x{h}: wrote outside the image"


_SPLIT_SNIPPET = """
import sys
sys.path.insert(0, {root!r})
import numpy as np, torch
from cuda_optical_flow_2_amd import lib, synth
L = lib.load()
st = torch.cuda.current_stream().cuda_stream
out = {{}}
for (w, h) in ((261, 35), (130, 131)):
    rng = np.random.default_rng(w)
    noise = np.repeat(rng.integers(0, 256, (h, w, 1), dtype=np.uint8), 3, axis=2)
    smooth = synth.to_3ch(synth.smooth_pair(w, h, 0, 0, seed=9)[1])
    for (ww, ss, sb) in ((7, 1.0, 5.0), (9, 2.0, 10.0), (13, 3.0, 40.0)):
        for name, img in (("noise", noise), ("smooth", smooth)):
            d_img = torch.from_numpy(img).cuda()
            d_out = torch.empty_like(d_img)
            lib.check(L.ofx_bilateral_3ch_fast(d_img.data_ptr(), d_img.data_ptr(), d_out.data_ptr(), w, h, ww, ww, ss, sb, st), "fast")
            out[f"{{w}}x{{h}}_{{ww}}_{{name}}"] = d_out.cpu().numpy()
np.savez(sys.argv[1], **out)
print("split ok")
"""


def test_fast_bilateral_filter_is_the_same_filter_however_its_columns_are_split(tmp_path):
    """bilateral_lut_kernel spreads a window's columns over three engines (LDS table, a gather out of the kernel-argument block,
    v_exp_f32); OFX_LUT_SPLIT picks the split per process.  Child processes with every split the library carries (0: all LDS,
    2: two computed, 10: one gathered, 12: the default) filter the same images; the outputs may differ from the default's by the
    rounding of a weight, never by more than one grey level -- the default is the one pinned to the oracle above."""
    import os
    import subprocess
    import sys as _sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for split in ("12", "0", "2", "10"):
        path = str(tmp_path / f"split{split}.npz")
        r = subprocess.run([_sys.executable, "-c", _SPLIT_SNIPPET.format(root=root), path], env=dict(os.environ, OFX_LUT_SPLIT=split),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "split ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
        got[split] = dict(np.load(path))
    for split in ("0", "2", "10"):
        for k, v in got["12"].items():
            d = np.abs(got[split][k].astype(np.int32) - v.astype(np.int32))
            assert d.max() <= 1, f"split {split}, {k}: off by {d.max()}"


@pytest.mark.parametrize("size", [(203, 77), (64, 4), (5, 3), (130, 131)])
def test_bilateral_filter_tiled_kernel_matches_oracle(gpu, cpu, oracle, size):
    """The tiled bilateral kernel (LDS neighbourhood, range table by signed difference, out-of-image taps as +0.0) is the
    reference's arithmetic in the reference's order: bit-exact against the oracle on sizes that are not multiples of the
    64 x 4 tile, windows 9x9 / 5x5 / 7x3 / 13x13, a colour image (three channels accumulated) and a grey one (one)."""
    w, h = size
    rng = np.random.default_rng(w * 1000 + h)
    colour = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    grey = oracle.grayscale_avg(colour)
    for (ww, wh, ss, sb) in ((9, 9, 2.0, 10.0), (5, 5, 1.5, 20.0), (7, 3, 1.0, 5.0), (13, 13, 3.0, 40.0)):
        assert_same(gpu.bilinear_filter(grey, grey, ww, wh, ss, sb), oracle.bilateral_3ch(grey, grey, ww, wh, ss, sb), f"grey {ww}x{wh} at {w}x{h}")
        assert_same(gpu.bilinear_filter(colour, grey, ww, wh, ss, sb), oracle.bilateral_3ch(colour, grey, ww, wh, ss, sb), f"colour {ww}x{wh} at {w}x{h}")
    assert_same(cpu.bilinear_filter_3ch(colour, grey, 9, 9, 2.0, 10.0), oracle.bilateral_3ch(colour, grey, 9, 9, 2.0, 10.0), "cpu:: twin")
