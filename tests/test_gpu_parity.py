"""GPU: parity of the HIP path (through the C ABI / the gpu:: drop-in surface) against the CPU oracle, the committed
golden fixtures, and -- at BASELINE.json's full sizes -- size-independent properties.

Tolerances: integer stages bit-exact; solve <= 1 float32 ulp with identical NaN/Inf mask (the HIP
path replays the reference's operation order in IEEE double, so it is in practice bit-identical and the tests
assert exact equality where the oracle uses the same exact window sums).
"""
import os

import numpy as np
import pytest

from conftest import assert_flow_close, assert_same
from cuda_optical_flow_2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cuda_optical_flow_2_amd import engine

    return engine


@pytest.fixture(scope="module")
def gpu():
    from cuda_optical_flow_2_amd.compat import GpuCompat

    return GpuCompat()


def _oracle_level(oracle, p, n, win, mode):
    p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
    h, w = p.shape
    fl = [np.zeros((h, w, 2), np.float32)]
    if mode == "compat_cpu":
        oracle.calc_optical_flow_cpu(p3, n3, fl, 0, 1, win)
    else:
        oracle.calc_opt_flow_gpu(p3, n3, fl, 0, 1, win, exact_sums=True)
    return fl[0]


def _oracle_sums(oracle, p, n, win, mode):
    _, _, _, sums = oracle.level_planes(synth.to_3ch(p), synth.to_3ch(n), win, 0 if mode == "compat_cpu" else 1, exact_sums=True)
    return sums


# ---- the fused level kernel ------------------------------------------------------------------------------------------

SHAPES = [(64, 48), (300, 37), (517, 64), (3, 5), (1, 1), (5, 1), (241, 9), (1000, 20)]


@pytest.mark.parametrize("mode", ["compat_cpu", "lk_float"])
@pytest.mark.parametrize("win", [3, 5, 7, 9, 15, 19, 23])
def test_level_sums_and_flow(eng, oracle, mode, win):
    for (w, h) in SHAPES:
        for gen in ("smooth", "random"):
            p, n = synth.smooth_pair(w, h, 0.6, -0.4) if gen == "smooth" else synth.random_pair(w, h, seed=w + h)
            want = _oracle_sums(oracle, p, n, win, mode)
            got = eng.lk_level(p, n, win, mode, want_sums=True)
            if mode == "compat_cpu":
                assert_same(got, want.astype(np.int64).astype(np.int32), f"sums {w}x{h} {gen} w{win}")
            else:  # the oracle's planes are float (rounded once); the kernel's are the exact integers
                assert_same(got.astype(np.float32), want.astype(np.float32), f"sums {w}x{h} {gen} w{win}")
            assert_same(eng.lk_level(p, n, win, mode), _oracle_level(oracle, p, n, win, mode), f"flow {w}x{h} {gen} w{win} {mode}")


def test_level_window_25_compat_only(eng, oracle):
    p, n = synth.random_pair(130, 40, seed=9)
    assert_same(eng.lk_level(p, n, 25, "compat_cpu"), _oracle_level(oracle, p, n, 25, "compat_cpu"), "w25 compat")
    from cuda_optical_flow_2_amd.lib import OfxError

    with pytest.raises(OfxError):
        eng.lk_level(p, n, 25, "lk_float")  # int32 window sums could overflow: refused, not wrapped
    with pytest.raises(OfxError):
        eng.lk_level(p, n, 8, "lk_float")   # even windows are not defined for the fused path


def test_level_flat_image_gives_nan(eng, oracle):
    """det == 0 everywhere: the reference divides by zero (no threshold) and so must we -- identical NaN mask."""
    p = np.full((40, 70), 77, np.uint8)
    for mode in ("compat_cpu", "lk_float"):
        got = eng.lk_level(p, p, 9, mode)
        want = _oracle_level(oracle, p, p, 9, mode)
        assert_same(got, want, "flat " + mode)
        assert np.isnan(got[10:30, 10:60]).all()


def test_level_against_float_order_oracle(eng, oracle):
    """vs the reference's row-major float accumulation (OptFlowGpu.cu:1569-1586): sums within 361*2^-24 of sum|ab|."""
    p, n = synth.smooth_pair(200, 120, 0.6, -0.4)
    _, _, _, ordered = oracle.level_planes(synth.to_3ch(p), synth.to_3ch(n), 19, 1, exact_sums=False)
    ix, iy, it, _ = oracle.level_planes(synth.to_3ch(p), synth.to_3ch(n), 19, 1, want_sums=False)
    got = eng.lk_level(p, n, 19, "lk_float", want_sums=True).astype(np.float64)
    pairs = ((ix, ix), (iy, iy), (ix, iy), (ix, it), (iy, it))
    for k, (a, b) in enumerate(pairs):
        bound = oracle.srm_1ch_f32(np.abs(a), np.abs(b), 19, 19, exact=True).astype(np.float64) * (361 * 2.0 ** -24)
        assert (np.abs(got[k] - ordered[k]) <= bound + 1e-6).all(), f"plane {k}"


@pytest.mark.parametrize("mode", ["compat_cpu", "lk_float"])
def test_level_row_sharding_is_bit_exact(eng, mode):
    """a level computed in row blocks (buffers = block + halo) equals the unsharded level bit for bit."""
    p, n = synth.random_pair(333, 90, seed=4)
    win, halo = 9, 9 // 2 + 1
    whole = eng.lk_level(p, n, win, mode)
    cuts = [0, 17, 18, 51, 90]
    for y0, y1 in zip(cuts[:-1], cuts[1:]):
        b0, b1 = max(0, y0 - halo), min(90, y1 + halo)
        part = eng.lk_level(p, n, win, mode, rows=(y0, y1), buf_rows=(b0, b1))
        assert_same(part, whole[y0:y1], f"rows [{y0},{y1})")
    from cuda_optical_flow_2_amd.lib import OfxError

    with pytest.raises(OfxError):  # halo too small: refused, never read out of the buffer
        eng.lk_level(p, n, win, mode, rows=(20, 40), buf_rows=(18, 42))


# ---- pyramid / shift / compose -----------------------------------------------------------------------------------------

def test_downsample(eng, oracle):
    for (w, h) in ((128, 96), (10, 6), (2, 2), (518, 34), (3840, 16)):
        img = synth.random_pair(w, h, seed=h)[0]
        assert_same(eng.downsample_1ch(img), oracle.downscale_gaussian(synth.to_3ch(img))[:, :, 0], f"downsample {w}x{h}")


def test_shift_golden_and_oracle(eng, oracle, golden):
    g = golden("shift")
    img = g["img"][:, :, 0]
    for i in range(len(g["uv"])):
        levels = [None, g[f"f1_{i}"], g[f"f2_{i}"]]
        uv = eng.shift_vector(levels, 0, 3)
        assert_same(eng.shift_1ch(img, uv), g[f"shift_{i}"][:, :, 0], f"shift case {i}")
    big = synth.random_pair(1001, 77, seed=8)[0]
    for uv in ((3.7, -2.2), (-1000.5, 0.0), (0.0, 76.5), (-0.999, -0.999)):
        fl = [None, np.array([[[uv[0] / 2, uv[1] / 2]]], np.float32)]
        got_uv = eng.shift_vector(fl, 0, 2)
        assert_same(eng.shift_1ch(big, got_uv), oracle.shift_back_pyramid(synth.to_3ch(big), 0, 2, fl)[:, :, 0], f"shift {uv}")


def test_compose_flow(eng, oracle):
    rng = np.random.default_rng(5)
    fl = [rng.normal(size=(48 >> k, 64 >> k, 2)).astype(np.float32) for k in range(3)]
    for level in (0, 1, 2):
        assert_same(eng.compose_flow(fl, 3, level), oracle.compose_flow(fl, 3, level), f"compose to level {level}")


# ---- whole pairs through the session -----------------------------------------------------------------------------------

def test_pair_golden_compat_cpu(eng, golden):
    """3-level 64x48 pipeline golden produced by the reference's own cpu::calc_optical_flow (window 9)."""
    g = golden("levels")
    got = eng.flow_pair(g["pair_prev"], g["pair_next"], 3, 9, "compat_cpu")
    for k in range(3):
        assert_flow_close(got[k], g[f"pair_flow_L{k}"], 1, f"pipeline flow L{k}")
        assert_same(got[k], g[f"pair_flow_L{k}"], f"pipeline flow L{k} (bit-exact)")
    for tag in ("smooth", "random"):
        one = eng.flow_pair(g[tag + "_prev"], g[tag + "_next"], 1, 9, "compat_cpu")[0]
        assert_same(one, g[tag + "_flow_single"], "single level " + tag)


@pytest.mark.parametrize("cfg", [(640, 480, 3, 5), (320, 240, 4, 19), (256, 192, 3, 9), (250, 186, 2, 7), (1002, 50, 2, 5), (6, 4, 2, 3),
                                 (516, 260, 3, 9)])  # the last four: widths that are not a multiple of 4 / 64 / 240 at some level
@pytest.mark.parametrize("mode", ["compat_cpu", "lk_float"])
def test_pair_vs_oracle(eng, oracle, cfg, mode):
    w, h, levels, win = cfg  # first entry is BASELINE config[0]: 640x480, 3 levels, 5x5
    p, n = synth.smooth_pair(w, h)
    got = eng.flow_pair(p, n, levels, win, mode)
    want, pp, npyr = oracle.flow_pair(synth.to_3ch(p), synth.to_3ch(n), levels, win, mode, exact_sums=True)
    for k in range(levels):
        assert_same(got[k], want[k], f"{mode} L{k}")


@pytest.mark.parametrize("cfg", [(640, 480, 3, 9, "lk_float"), (320, 700, 2, 15, "lk_float"), (252, 400, 3, 5, "compat_cpu")])
def test_tall_strips_refresh_the_row_map(eng, oracle, cfg, monkeypatch):
    """The march keeps the reference's row map (int)((float)y + v) for 64 rows at a time in a lane table and refreshes it
    when the entering row runs off its end (lk_body.h).  Small frames are normally cut into 8-row strips, which never get
    there: OFX_LK_MIN_STRIP forces strips as tall as the level, so every wave refreshes several times -- plain launch and
    stream kernel against the oracle."""
    w, h, levels, win, mode = cfg
    monkeypatch.setenv("OFX_LK_MIN_STRIP", "1000")
    frames = [synth.smooth_pair(w, h, 1.7 * i, -2.3 * i, seed=11)[1] for i in range(5)]
    want = [oracle.flow_pair(synth.to_3ch(frames[i - 1]), synth.to_3ch(frames[i]), levels, win, mode, exact_sums=True)[0] for i in range(1, 5)]
    for i in range(1, 5):
        got = eng.flow_pair(frames[i - 1], frames[i], levels, win, mode)
        for k in range(levels):
            assert_same(got[k], want[i - 1][k], f"plain pair {i} L{k}")
    import torch

    s = eng.Session(w, h, levels, win, mode, stream_batch=2)
    s.stream_begin()
    d = [torch.from_numpy(f).cuda() for f in frames]
    seen = {}
    def snap(done):
        for p in (done - 1, done):
            if p >= 1 and p not in seen:
                seen[p] = [s.flow_of(p, k)[0].cpu().numpy() for k in range(levels)]
    for f in d:
        snap(s.stream_submit(f))
    while True:
        done = s.stream_drain()
        if done == -2:
            break
        snap(done)
    s.close()
    assert sorted(seen) == [1, 2, 3, 4]
    for p in seen:
        for k in range(levels):
            assert_same(seen[p][k], want[p - 1][k], f"stream pair {p} L{k}")


def test_sharded_session_reports_a_shift_beyond_its_halo(eng):
    """A rank's buffers hold `margin` rows of slack for the reference's global shift.  A pair whose pixel-0 flow is wild
    (found by tools/fuzz_stream.py: a degenerate corner gives a level-1 shift of -46 rows) reaches past it; the rows next to
    the shard's edge are then not the reference's, and the corner stage must say so: bit 8 + level of the status word."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win, R = 1104, 128, 4, 3, 3
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.1 * i, -0.7 * i, seed=251)[1]).cuda() for i in (24, 25, 24, 25)]
    plain = eng.Session(w, h, L, win, "lk_float")
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    plain.set_frame_device(frames[1]); plain.build_pyramid(); plain.run_flow()
    torch.cuda.synchronize()
    v1 = float(plain.uv(1).cpu()[1])
    plain.close()
    assert abs(v1) > 40, "this pair is expected to have a wild level-1 shift"
    ranks = [eng.Session(w, h, L, win, "lk_float", shard=ShardPlan(w, h, L, win, r, R), local_corner=True, strict=False) for r in range(R)]
    for s in ranks:
        s.stream_begin()
        for f in frames:
            s.stream_submit(f)
        while s.stream_drain() != -2:
            pass
    torch.cuda.synchronize()
    status = [s.corner_status() for s in ranks]
    for s in ranks:
        s.close()
    # (rank 0 holds the top rows of every level, which is where an upward shift lands: the ranks below it must flag level 1)
    assert any((st >> 8) & 0b10 for st in status[1:]), [hex(st) for st in status]


def test_pair_at_a_time_borrowed_frames(eng):
    """ofx_params.borrow_frames on the pair-at-a-time path: ofx_session_set_frame_device only remembers the buffer, the
    pyramid, corner and LK launches read level 0 in place (no copy launch).  Same bits as the copying session; a frame whose
    pitch is not the session's is refused."""
    import torch

    w, h, L, win = 640, 480, 4, 9
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.3 * i, 0.8 * i, seed=78)[1]).cuda() for i in range(5)]
    a = eng.Session(w, h, L, win, "lk_float")
    b = eng.Session(w, h, L, win, "lk_float", borrow_frames=True)
    for s in (a, b):
        s.set_frame_device(frames[0]); s.build_pyramid(); s.swap()
    for i in range(1, 5):
        for s in (a, b):
            s.set_frame_device(frames[i]); s.build_pyramid(); s.run_flow()
        torch.cuda.synchronize()
        for k in range(L):
            assert_same(b.flow_host(k), a.flow_host(k), f"pair {i} level {k}")
        a.swap(); b.swap()
    # a host frame on a borrowing session goes into the session's own plane again
    b.set_frame_host(frames[0].cpu().numpy()); b.build_pyramid(); b.run_flow()
    a.set_frame_device(frames[0]); a.build_pyramid(); a.run_flow()
    torch.cuda.synchronize()
    assert_same(b.flow_host(0), a.flow_host(0), "host frame after borrowed frames")
    a.close(); b.close()
    c = eng.Session(250, 186, 2, 7, "lk_float", borrow_frames=True)
    with pytest.raises(RuntimeError, match="pitch"):
        c.set_frame_device(torch.zeros((186, 250), dtype=torch.uint8, device="cuda"))
    c.close()


def test_stream_submit_frames_equals_single_submits(eng):
    """ofx_session_stream_submit_frames(n frames) == n calls of ofx_session_stream_submit: same pairs reported, same bits;
    group sizes that do and do not line up with the frames per launch."""
    import torch

    w, h, L, win = 320, 240, 3, 7
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.5 * i, 0.5 * i, seed=3)[1]).cuda() for i in range(13)]

    def run(group_sizes, batch):
        s = eng.Session(w, h, L, win, "lk_float", stream_batch=batch)
        s.stream_begin()
        out, seen, i = {}, 0, 0
        def snap(done):
            nonlocal seen
            for p in range(max(seen + 1, done - batch + 1), done + 1):
                out[p] = [s.flow_of(p, k)[0].cpu().numpy() for k in range(L)]
            seen = max(seen, done)
        for g in group_sizes:
            done = s.stream_submit_frames(frames[i:i + g]) if g > 1 else s.stream_submit(frames[i])
            i += g
            if done >= 1:
                snap(done)
        assert i == len(frames)
        while True:
            done = s.stream_drain()
            if done == -2:
                break
            if done >= 1:
                snap(done)
        s.close()
        return out

    for batch in (1, 4):
        ref = run([1] * 13, batch)
        assert sorted(ref) == list(range(1, 13))
        for sizes in ([4, 4, 4, 1], [3, 5, 2, 3], [13]):
            got = run(sizes, batch)
            # a group larger than a tick completes several ticks in one call: only the newest `batch` pairs are still readable
            assert set(got) <= set(ref) and max(got) == 12
            for p in got:
                for k in range(L):
                    assert_same(got[p][k], ref[p][k], f"batch {batch} groups {sizes} pair {p} L{k}")


def test_session_planes_and_streaming(eng, oracle):
    """Pyramid planes equal the oracle's, and prev/next swap keeps the previous pyramid (main.cu:270-272)."""
    import torch

    w, h, L = 192, 128, 3
    frames = [synth.smooth_pair(w, h, dx, 0.5 * dx, seed=77)[1] for dx in (0.0, 1.0, 2.5)]
    s = eng.Session(w, h, L, 7, "lk_float")
    s.push_frame_host(frames[0])
    for i in (1, 2):
        s.set_frame_host(frames[i])
        s.build_pyramid()
        s.run_flow()
        torch.cuda.synchronize()
        want, pp, npyr = oracle.flow_pair(synth.to_3ch(frames[i - 1]), synth.to_3ch(frames[i]), L, 7, "lk_float", exact_sums=True)
        for k in range(L):
            plane, g = s.plane(1, k)
            assert_same(plane[:, : g.w].cpu().numpy(), npyr[k][:, :, 0], f"next pyramid L{k}")
            assert_same(s.flow_host(k), want[k], f"frame {i} flow L{k}")
        s.swap()
    s.close()


@pytest.mark.parametrize("mode", ["compat_cpu", "lk_float"])
def test_corner_kernel_path_equals_sequential_path(eng, mode):
    """run_flow = corner kernel + one multi-level shift launch + one multi-level LK launch; it must reproduce the
    level-by-level sequence of main.cu:256-262 bit for bit (flows AND shift vectors)."""
    import torch

    for (w, h, L, win, gen) in ((320, 240, 4, 9, "smooth"), (256, 192, 3, 5, "random"), (64, 48, 3, 19, "smooth"), (16, 8, 2, 3, "random")):
        p, n = synth.smooth_pair(w, h) if gen == "smooth" else synth.random_pair(w, h, seed=11)
        s = eng.Session(w, h, L, win, mode)
        s.push_frame_host(p)
        s.set_frame_host(n)
        s.build_pyramid()
        s.run_flow_sequential()
        torch.cuda.synchronize()
        seq = [s.flow_host(k) for k in range(L)]
        uv_seq = [s.uv(k).cpu().numpy().copy() for k in range(L - 1)]
        for k in range(L):
            s.flow(k)[0].zero_()
        for k in range(L - 1):
            s.uv(k).zero_()
        s.run_flow()
        torch.cuda.synchronize()
        for k in range(L):
            assert_same(s.flow_host(k), seq[k], f"{w}x{h} {mode} L{k}")
        for k in range(L - 1):
            assert_same(s.uv(k).cpu().numpy(), uv_seq[k], f"{w}x{h} {mode} uv L{k}")
        s.close()


@pytest.mark.parametrize("mode", ["lk_float", "compat_cpu"])
def test_sharded_sessions_on_one_device(eng, mode):
    """SURVEY 8e verification step: R logical ranks (sharded HIP sessions built from parallel.ShardPlan) on one device
    reproduce the unsharded result bit for bit.  Rank 0's shift vectors are handed to the others with a device copy --
    the place the RCCL broadcast takes in a real multi-GPU run (tests/test_parallel.py covers that under gloo)."""
    import torch
    from cuda_optical_flow_2_amd.parallel import HipBackend, ShardPlan

    w, h, L, win, R = 640, 480, 4, 9, 4
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.5 * i, 0.75 * i, seed=3)[1]).cuda() for i in range(3)]
    whole = eng.Session(w, h, L, win, mode)
    ranks = [HipBackend(ShardPlan(w, h, L, win, r, R), mode, 0) for r in range(R)]
    for b in [whole] + [r.session for r in ranks]:
        b.set_frame_device(frames[0])
        b.build_pyramid()
        b.swap()
    for i in (1, 2):
        whole.set_frame_device(frames[i])
        whole.build_pyramid()
        whole.run_flow()
        for r in ranks:
            r.load_frame(frames[i])
            r.build_pyramid()
        ranks[0].corner_flows()
        for r in ranks[1:]:
            r.uv_all.copy_(ranks[0].uv_all)
        for r in ranks:
            r.run_levels()
        torch.cuda.synchronize()
        for k in range(L):
            got = torch.cat([r.flow(k) for r in ranks], dim=0).cpu().numpy()
            assert_same(got, whole.flow_host(k), f"{mode} frame {i} level {k}")
        whole.swap()
        for r in ranks:
            r.swap()
    whole.close()
    for r in ranks:
        r.session.close()


@pytest.mark.parametrize("mode", ["lk_float", "compat_cpu"])
def test_halo_exchange_sessions_on_one_device(eng, mode):
    """halo_mode="exchange" on HIP sessions: every logical rank is handed ONLY its own rows of a frame (the rest is poison),
    downsamples only its own rows (ofx_session_downsample_level with comp == own) and gets the halo rows of every level
    from its neighbours' planes -- device copies here, where ShardedFlow.exchange_halos posts the RCCL send/recv pairs
    (tests/test_parallel.py runs that code under gloo).  Bit-exact against the unsharded session."""
    import torch
    from cuda_optical_flow_2_amd.parallel import HipBackend, ShardPlan

    w, h, L, win, R = 640, 960, 4, 9, 4   # 120 coarse rows: 30 per rank >= halo (4 + 1 + 8)
    frames = [synth.smooth_pair(w, h, 1.5 * i, 0.75 * i, seed=3)[1] for i in range(3)]
    whole = eng.Session(w, h, L, win, mode)
    plans = [ShardPlan(w, h, L, win, r, R, 8, "exchange") for r in range(R)]
    ranks = [HipBackend(pl, mode, 0) for pl in plans]

    def load_and_build(i):
        for pl, b in zip(plans, ranks):
            seen = np.full_like(frames[i], 0xEE)
            seen[pl.own[0][0]: pl.own[0][1]] = frames[i][pl.own[0][0]: pl.own[0][1]]
            b.load_frame(torch.from_numpy(seen).cuda())
        for k in range(L):
            if k > 0:
                for b in ranks:
                    b.downsample_level(k)
            views = [b.next_plane(k) for b in ranks]
            wk = w >> k
            for r, (pl, (t, base)) in enumerate(zip(plans, views)):   # halos from the neighbours' OWN rows
                (o0, o1), (b0, b1) = pl.own[k], pl.buf[k]
                if r > 0:
                    src, sbase = views[r - 1]
                    t[b0 - base: o0 - base, :wk] = src[b0 - sbase: o0 - sbase, :wk]
                if r + 1 < R:
                    src, sbase = views[r + 1]
                    t[o1 - base: b1 - base, :wk] = src[o1 - sbase: b1 - sbase, :wk]

    whole.set_frame_device(torch.from_numpy(frames[0]).cuda()); whole.build_pyramid(); whole.swap()
    load_and_build(0)
    for b in ranks:
        b.swap()
    for i in (1, 2):
        whole.set_frame_device(torch.from_numpy(frames[i]).cuda()); whole.build_pyramid(); whole.run_flow()
        load_and_build(i)
        ranks[0].corner_flows()
        for b in ranks[1:]:
            b.uv_all.copy_(ranks[0].uv_all)
        for b in ranks:
            b.run_levels()
        torch.cuda.synchronize()
        for k in range(L):
            got = torch.cat([b.flow(k) for b in ranks], dim=0).cpu().numpy()
            assert_same(got, whole.flow_host(k), f"{mode} frame {i} level {k}")
        whole.swap()
        for b in ranks:
            b.swap()
    whole.close()
    for b in ranks:
        b.session.close()


@pytest.mark.parametrize("mode", ["lk_float", "compat_cpu"])
def test_pipelined_submit_equals_plain_sequence(eng, mode):
    """submit_device (staging on the aux stream under the previous LK launch, three rotating image sets) must give, for
    EVERY pair of a back-to-back stream, the bits of set_frame/build_pyramid/run_flow/swap.  Flow snapshots are taken with
    stream-ordered device copies so that the pairs really are in flight together."""
    import torch

    w, h, L, win, nf = 1280, 720, 4, 9, 6   # large enough that an LK launch outlasts the next pair's staging
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.7 * i, -0.9 * i, seed=21)[1]).cuda() for i in range(nf)]
    plain = eng.Session(w, h, L, win, mode)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = []
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want.append([plain.flow_host(k) for k in range(L)])
        plain.swap()
    plain.close()

    piped = eng.Session(w, h, L, win, mode)
    piped.set_frame_device(frames[0]); piped.build_pyramid(); piped.swap()
    torch.cuda.synchronize()
    snaps = []
    for i in range(1, nf):
        piped.submit_device(frames[i])
        snaps.append([piped.flow(k)[0].clone() for k in range(L)])   # ordered after LK(i) on the current stream
    torch.cuda.synchronize()
    for i, (got, ref) in enumerate(zip(snaps, want)):
        for k in range(L):
            assert_same(got[k].cpu().numpy(), ref[k], f"{mode} pair {i + 1} level {k}")
    piped.close()


@pytest.mark.parametrize("cfg", [(512, 384, 4, 9, "lk_float"), (512, 384, 4, 9, "compat_cpu"), (320, 200, 3, 15, "lk_float"), (64, 32, 2, 3, "lk_float"),
                                 (250, 186, 2, 7, "lk_float"), (516, 260, 3, 9, "compat_cpu"), (1280, 768, 7, 5, "lk_float")])
def test_stream_pipeline_equals_plain_sequence(eng, cfg):
    """The one-launch-per-frame stream pipeline (pyramid | corner | shift | LK of four consecutive pairs side by side in one
    grid) must reproduce, pair by pair, the bits of the plain sequence; pair p's flow appears with frame p+2."""
    import torch

    w, h, L, win, mode = cfg
    nf = 9
    # frames live in padded buffers (row pitch != width; the stream path needs a pitch that is a multiple of 4, dirty padding
    # bytes must not leak into the result)
    pitch = (w + 3) // 4 * 4 + 4
    def padded(a):
        buf = torch.full((h, pitch), 0xA5, dtype=torch.uint8, device="cuda")
        buf[:, :w] = torch.from_numpy(a).cuda()
        return buf[:, :w]
    frames = [padded(synth.smooth_pair(w, h, 1.3 * i, -0.7 * i, seed=31)[1]) for i in range(nf)]
    plain = eng.Session(w, h, L, win, mode)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want[i] = [plain.flow_host(k) for k in range(L)]
        plain.swap()
    plain.close()

    s = eng.Session(w, h, L, win, mode)
    s.stream_begin()
    got = {}
    for i in range(nf):
        done = s.stream_submit(frames[i])
        assert done == (i - 2 if i - 2 >= 1 else -1)
        if done >= 1:
            got[done] = [s.flow(k)[0].clone() for k in range(L)]   # stream-ordered snapshot, launches stay in flight
    while True:
        done = s.stream_drain()
        if done == -2:
            break
        if done >= 1:
            got[done] = [s.flow(k)[0].clone() for k in range(L)]
    torch.cuda.synchronize()
    assert sorted(got) == list(range(1, nf))
    for p in range(1, nf):
        for k in range(L):
            assert_same(got[p][k].cpu().numpy(), want[p][k], f"{mode} pair {p} level {k}")
    s.close()


@pytest.mark.parametrize("cfg", [(640, 480, 4, 9, "lk_float", 4), (640, 480, 4, 9, "compat_cpu", 3), (1920, 1088, 5, 7, "lk_float", 8),
                                 (768, 512, 3, 15, "lk_float", 2)])
def test_sharded_stream_sessions_with_local_corner(eng, cfg):
    """Row-sharded sessions running the ONE-LAUNCH stream pipeline with the corner flows computed locally from each
    frame's top-left patch (ofx_params.local_corner): R logical ranks on one device, nothing passed between them, must
    reproduce the unsharded plain sequence bit for bit for every pair, and no rank may report that the shift left its
    patch."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win, mode, R = cfg
    nf = 6
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.4 * i, 0.8 * i, seed=17)[1]).cuda() for i in range(nf)]
    plain = eng.Session(w, h, L, win, mode)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want[i] = [plain.flow_host(k) for k in range(L)]
        plain.swap()
    plain.close()

    plans = [ShardPlan(w, h, L, win, r, R) for r in range(R)]
    ranks = [eng.Session(w, h, L, win, mode, shard=pl, local_corner=True) for pl in plans]
    got = {}
    for s in ranks:
        s.stream_begin()
    def snap(done):
        if done >= 1:
            got[done] = [[s.flow(k)[0].clone() for k in range(L)] for s in ranks]
    for i in range(nf):
        dones = [s.stream_submit(frames[i]) for s in ranks]
        assert len(set(dones)) == 1
        snap(dones[0])
    while True:
        dones = [s.stream_drain() for s in ranks]
        assert len(set(dones)) == 1
        if dones[0] == -2:
            break
        snap(dones[0])
    torch.cuda.synchronize()
    assert sorted(got) == list(range(1, nf))
    for p in range(1, nf):
        for k in range(L):
            full = torch.cat([got[p][r][k] for r in range(R)], dim=0).cpu().numpy()
            assert_same(full, want[p][k], f"{mode} pair {p} level {k}")
    for s in ranks:
        assert s.corner_status() == 0
        s.close()


@pytest.mark.parametrize("cfg", [(1280, 720, 4, 9, 4, 2, False), (1920, 1088, 5, 7, 8, 4, True), (640, 480, 3, 5, 3, 1, True)])
def test_sharded_stream_sessions_read_only_their_rows_and_the_patch(eng, cfg):
    """ShardedFlow's "stream_exchange" mode hands a rank frame buffers in which only its level-0 buffer rows (own + halo) and
    the frame's top-left patch were ever written (parallel.py, assemble_frames: that is all that crosses ranks).  Here R
    logical ranks on one device get exactly such buffers -- everything else poisoned -- through the stream pipeline (copied and
    borrowed frames): every pair must equal the unsharded plain sequence bit for bit, i.e. the session reads nothing else."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win, R, B, borrow = cfg
    nf = 3 * B + 2
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.3 * i, -0.7 * i, seed=23)[1]).cuda() for i in range(nf)]
    want = _plain_sequence(eng, frames, w, h, L, win)
    plans = [ShardPlan(w, h, L, win, r, R, halo_mode="stream_exchange") for r in range(R)]
    got = {}
    for r, pl in enumerate(plans):
        pw, ph = pl.patch_wh(0)
        b0, b1 = pl.buf[0]
        bufs = []
        for f in frames:
            t = torch.full((h, w), 0xEE, dtype=torch.uint8, device="cuda")
            t[b0:b1] = f[b0:b1]
            t[:ph, :pw] = f[:ph, :pw]
            bufs.append(t)
        s = eng.Session(w, h, L, win, "lk_float", shard=pl, local_corner=True, stream_batch=B, borrow_frames=borrow, frames_partial=True)
        got[r] = _stream_all_pairs(s, bufs, L, B)
        assert s.corner_status() == 0
        s.close()
    for p in range(1, nf):
        for k in range(L):
            full = np.concatenate([got[r][p][k] for r in range(R)], axis=0)
            assert_same(full, want[p][k], f"{R} ranks, pair {p} level {k}")


@pytest.mark.parametrize("iters", [1, 3])
def test_own_rows_ticks_with_the_exchange_on_a_side_stream_equal_the_plain_sequence(eng, iters):
    """ShardedFlow.stream_submit_own_rows(overlap=True): a tick's assembly (own rows -> ring buffers, the exchange) runs on a side
    stream one tick ahead of the launch that reads it, events in between.  One rank has nothing to exchange, but the ordering is
    the same: every pair of twelve ticks' worth of frames -- the ring of assembled buffers comes round more than twice -- must equal
    the plain sequence bit for bit, with the returned pair numbers one tick late and stream_drain() flushing the waiting tick."""
    import torch
    from cuda_optical_flow_2_amd import parallel

    w, h, L, win, B = 1280, 720, 4, 9, 2
    nf = 12 * B
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.1 * i, -0.6 * i, seed=41)[1]).cuda() for i in range(nf)]
    plain = eng.Session(w, h, L, win, "lk_float", iters=iters)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        want[i] = [plain.flow(k)[0].clone() for k in range(L)]
        plain.swap()
    plain.close()
    drv = parallel.ShardedFlow(w, h, L, win, "lk_float", 0, 1, device=0, corner="local", stream_batch=B, halo_mode="stream_exchange",
                               borrow_frames=True, iters=iters)
    s = drv.session
    drv.stream_begin()
    got, returned = {}, []

    def snap(done):
        for p in range(max(1, done - B + 1), done + 1):
            if p not in got:
                got[p] = [s.flow_of(p, k)[0].clone() for k in range(L)]
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):   # (a caller's stream that is not the default one)
        for t in range(nf // B):
            rows = torch.stack(frames[t * B:(t + 1) * B])   # one rank: its own rows are the whole frame
            done = drv.stream_submit_own_rows(rows, overlap=True)
            del rows                                        # the driver must keep what its side stream still reads
            returned.append(done)
            if done >= 1:
                snap(done)
        while True:
            done = drv.stream_drain()
            if done == -2:
                break
            if done >= 1:
                snap(done)
    torch.cuda.synchronize()
    assert returned[0] == -1 and returned[1] == -1, returned   # nothing is launched by the first call; the second launches tick 0
    for p in range(1, nf):
        for k in range(L):
            a, b = got[p][k], want[p][k]
            assert bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all().item()), (iters, p, k)
    assert drv.corner_status() == 0
    s.close()


def test_local_corner_reports_a_shift_that_leaves_the_patch(eng):
    """ofx_session_corner_status must say exactly when a corner shift needed pixels the patch does not hold.  Pixel 0's
    flow is normally tiny (the zero border dominates its gradients), so the frames are a dark-cornered ramp whose
    brightness steps by d between frames: that yields corner shifts of several pixels, against a deliberately minimal
    patch.  The expected bits are recomputed here from the shift vectors the session publishes."""
    import torch

    w, h, L, win, patch = 640, 480, 3, 3, 12
    R = win // 2
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = xx / 2 + yy / 3 + xx * yy / 64
    seen_miss = False
    for d in (10, 20, 40, 80):
        a = np.clip(np.floor(base), 0, 255).astype(np.uint8)
        b = np.clip(np.floor(base + d), 0, 255).astype(np.uint8)
        s = eng.Session(w, h, L, win, "lk_float", local_corner=True, patch_size=patch, strict=False)   # (copied frames: no repair)
        s.stream_begin()
        expected = 0
        for i, f in enumerate([a, b, a, b]):
            s.stream_submit(torch.from_numpy(f).cuda())
            torch.cuda.synchronize()
            if i >= 2:   # the corner stage of tick i handled pair i-1, whose vectors sit in slot (i-1) & 1
                uvp = s.uv(0).data_ptr()
                both = eng.DeviceView(uvp, (2 * 12 * 2,), "<f4").tensor().cpu().numpy().reshape(2, 12, 2)
                # uv(0) points at the session's current slot, which the stream pipeline does not advance: slot 0 is first
                uv = both[(i - 1) & 1]
                for k in range(L - 1):
                    wk, hk, pk = w >> k, h >> k, patch >> k
                    u, v = np.float32(uv[k][0]), np.float32(uv[k][1])
                    for y in range(-1, R + 2):
                        for x in range(-1, R + 2):
                            if not (0 <= x < wk and 0 <= y < hk):
                                continue
                            tx, ty = np.float32(x) + u, np.float32(y) + v
                            if tx > -1 and tx < wk and ty > -1 and ty < hk and (int(tx) >= pk or int(ty) >= pk):
                                expected |= 1 << k
        while s.stream_drain() != -2:
            pass
        torch.cuda.synchronize()
        got = s.corner_status()
        assert got == expected, (d, got, expected)
        assert s.corner_status() == 0   # reading clears it
        seen_miss = seen_miss or got != 0
        s.close()
    assert seen_miss, "no brightness step drove the corner shift out of the patch: the test lost its subject"


def test_partial_frames_report_a_shift_that_leaves_the_patch_instead_of_repairing_it(eng):
    """a rank of ShardedFlow's "stream_exchange" mode holds its plan's rows and the top-left patch of a frame,
    nothing else.  With borrowed frames the session used to REPAIR a corner shift that leaves the patch by rebuilding the patch
    pyramid around the target from the whole next frame -- here from bytes that never arrived -- and said nothing.  With
    ofx_params.frames_partial the repair is off and the pair raises bit k, exactly the bits a session with copied frames (which
    cannot repair either) raises on the whole frames; the poison outside the rows and the patch must not change them."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win, patch = 640, 480, 3, 3, 12
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    base = xx / 2 + yy / 3 + xx * yy / 64
    seen_miss = False
    for d in (20, 80):
        a = np.clip(np.floor(base), 0, 255).astype(np.uint8)
        b = np.clip(np.floor(base + d), 0, 255).astype(np.uint8)
        seq = [torch.from_numpy(f).cuda() for f in (a, b, a, b)]
        ref = eng.Session(w, h, L, win, "lk_float", local_corner=True, patch_size=patch, strict=False)   # copied whole frames: reports, cannot repair
        _stream_all_pairs(ref, seq, L, 1)
        want_bits = ref.corner_status() & ((1 << L) - 1)
        ref.close()
        for r in range(2):
            pl = ShardPlan(w, h, L, win, r, 2, halo_mode="stream_exchange")
            pw, ph = pl.patch_wh(patch)
            b0, b1 = pl.buf[0]
            bufs = []
            for f in seq * 2:   # (distinct buffers: borrowed frames stay in use for three ticks)
                t = torch.full((h, w), 0xEE, dtype=torch.uint8, device="cuda")
                t[b0:b1] = f[b0:b1]
                t[:ph, :pw] = f[:ph, :pw]
                bufs.append(t)
            s = eng.Session(w, h, L, win, "lk_float", shard=pl, local_corner=True, patch_size=patch, borrow_frames=True, frames_partial=True, strict=False)
            _stream_all_pairs(s, bufs[:4], L, 1)
            got_bits = s.corner_status()
            s.close()
            assert got_bits & ((1 << L) - 1) == want_bits, (d, r, hex(got_bits), hex(want_bits))
        seen_miss = seen_miss or want_bits != 0
    assert seen_miss, "no brightness step drove the corner shift out of the patch: the test lost its subject"


def _plain_sequence(eng, frames, w, h, L, win, mode="lk_float"):
    """flows of every pair of `frames` through the plain pair-at-a-time session (the sequence the oracle tests pin)"""
    import torch

    plain = eng.Session(w, h, L, win, mode)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, len(frames)):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want[i] = [plain.flow_host(k) for k in range(L)]
        plain.swap()
    plain.close()
    return want


def _stream_all_pairs(s, frames, L, B):
    """every pair of `frames` through the stream pipeline of session s; returns {pair: [flow per level]} (host copies)"""
    import torch

    got, seen = {}, 0

    def snap(done):
        nonlocal seen
        if done >= 1:
            for p in range(max(seen + 1, done - B + 1), done + 1):
                got[p] = [s.flow_of(p, k)[0].cpu().numpy() for k in range(L)]
            seen = done

    s.stream_begin()
    for f in frames:
        snap(s.stream_submit(f))
    while True:
        done = s.stream_drain()
        if done == -2:
            break
        snap(done)
    torch.cuda.synchronize()
    return got


@pytest.mark.parametrize("cfg", [(1280, 720, 4, 9, "lk_float", 4, True), (1000, 564, 3, 7, "lk_float", 2, False), (640, 480, 3, 5, "compat_cpu", 4, True),
                                 (1920, 1080, 4, 15, "lk_float_fast", 2, True)])
@pytest.mark.parametrize("hint", [1, -1])
def test_deep_fetch_hint_changes_no_bit(eng, cfg, hint):
    """ofx_params.deep_fetch (ABI v10) says where the caller's frames come from -- +1: cold, the tick's LK stage fetches its rows two
    steps ahead straight into LDS; -1: warm, one step ahead -- and is a hint about speed only: every pair of a stream carries the
    bits of the plain pair-at-a-time sequence (the one the oracle tests pin) under either value, in two stages and in three."""
    import torch

    w, h, L, win, mode, B, two_stage = cfg
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.5 * i, -0.75 * i, seed=77)[1]).cuda() for i in range(3 * B + 2)]
    want = _plain_sequence(eng, frames, w, h, L, win, mode)
    s = eng.Session(w, h, L, win, mode, stream_batch=B, borrow_frames=True, two_stage=two_stage, deep_fetch=hint)
    got = _stream_all_pairs(s, frames, L, B)
    assert s.corner_status() == 0
    s.close()
    assert sorted(got) == list(range(1, len(frames)))
    for p in got:
        for k in range(L):
            assert_same(got[p][k], want[p][k], f"deep_fetch {hint:+d}: {mode} pair {p} level {k}")
    with pytest.raises(Exception):
        eng.Session(w, h, L, win, mode, deep_fetch=2)


def test_single_level_session_with_iterations_on_borrowed_frames(eng):
    """The fused warp of a refinement iteration fetches a tap's dword at the tap's own byte (lk_body_warp.h), so its source must be
    followed by three readable bytes.  Every plane of a session is; the one warp source that would be a caller's buffer is level 0
    of a single-level session on borrowed frames (pair at a time: the stream pipeline needs two levels), which therefore COPIES its
    frames whatever borrow_frames says (include/ofx.h).  The frames here are exactly-sized tensors, and the flow of noise frames sends
    taps to the last rows' last columns; every pair must carry the bits of the session that copies."""
    import torch

    w, h, L, win, iters = 640, 360, 1, 9, 4
    frames = [torch.from_numpy(synth.random_pair(w, h, seed=300 + i)[0]).cuda() for i in range(4)]
    got = {}
    for borrow in (False, True):
        s = eng.Session(w, h, L, win, "lk_float", iters=iters, borrow_frames=borrow)
        s.set_frame_device(frames[0]); s.build_pyramid(); s.swap()
        got[borrow] = []
        for i in range(1, len(frames)):
            s.set_frame_device(frames[i]); s.build_pyramid(); s.run_flow()
            torch.cuda.synchronize()
            got[borrow].append(s.flow_host(0))
            s.swap()
        s.close()
    for i, (a, b) in enumerate(zip(got[False], got[True])):
        assert_same(b, a, f"single level, {iters} iterations, borrowed frames, pair {i + 1}")


@pytest.mark.parametrize("kind", ["two_stage", "local_corner"])
@pytest.mark.parametrize("batch", [1, 2, 4])
def test_corner_shift_that_leaves_the_patch_is_repaired(eng, oracle, kind, batch, monkeypatch):
    """cpu::shift_back_pyramid defines the shift for EVERY input (OptFlowCPU.cpp:255-273).  The fast stream paths form the shift
    vectors from a small top-left patch of each frame; when a shifted corner leaves it, the corner block rebuilds the next
    frame's patch pyramid around the target (ofx_corner_stage.d_patch_reloc, sessions on borrowed frames) and the pair is the
    reference's all the same.  Pixel 0's flow saturates at a few pixels per level on anything but adversarial input (the
    zero border dominates its gradients), always inside a patch of the automatic size, so the test narrows what the chain may
    read of its patch planes to 16 level-0 pixels (OFX_DEBUG_CORNER_EXTENT, a hook for this test) and feeds pairs that
    darken over a shallow ramp: shifts of 4 / 13 / 30 pixels at levels 2 / 1 / 0.  Every pair must carry the bits of the plain
    sequence -- whose first pair is tied to the oracle here -- with status word 0, and the pairs' own words must say that
    relocated planes were used."""
    import torch

    w, h, L, win = 1024, 768, 4, 5
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    tex = synth.smooth_pair(w, h, 0, 0, seed=3)[1].astype(np.float64)
    host = []
    for g, d in [(0.25, 50), (0.25, 0), (0.5, 25), (0.5, 0), (0.125, 12), (0.125, 0), (1.0, 50), (1.0, 3), (0.25, 25)]:
        base = xx * g + yy * g * 0.7 + 0.15 * tex
        host.append(np.clip(np.floor(base + d), 0, 255).astype(np.uint8))   # (darkening pairs: positive shifts, into the image)
    frames = [torch.from_numpy(f).cuda() for f in host]
    want = _plain_sequence(eng, frames, w, h, L, win)
    fl, _, _ = oracle.flow_pair(synth.to_3ch(host[0]), synth.to_3ch(host[1]), L, win, "lk_float", exact_sums=True)
    for k in range(L):
        assert_same(want[1][k], fl[k], f"plain sequence vs oracle, level {k}")
    # the subject of the test: the reference's shift of level 1 (from pixel 0 of the coarser flows) is beyond the narrowed planes
    u1 = sum(np.float32(1 << (j - 1)) * fl[j][0, 0, 0] for j in range(L - 1, 1, -1))
    assert u1 > 8, f"level-1 shift {u1}: the frames no longer push the corner out of the narrowed patch"
    monkeypatch.setenv("OFX_DEBUG_CORNER_EXTENT", "16")
    if kind == "two_stage":
        s = eng.Session(w, h, L, win, "lk_float", stream_batch=batch, borrow_frames=True, two_stage=True)
    else:
        s = eng.Session(w, h, L, win, "lk_float", stream_batch=batch, borrow_frames=True, local_corner=True)
    monkeypatch.delenv("OFX_DEBUG_CORNER_EXTENT")
    got = _stream_all_pairs(s, frames, L, batch)
    assert sorted(got) == list(range(1, len(frames)))
    for p in got:
        for k in range(L):
            assert_same(got[p][k], want[p][k], f"{kind} batch {batch} pair {p} level {k}")
    assert s.corner_status() == 0
    words = [s.pair_status(p) for p in range(max(1, len(frames) - 2 * batch), len(frames))]
    assert all((wd & 0xFFFFFF) == 0 for wd in words), words
    assert any(wd & s.STATUS_REPAIRED for wd in words), f"no pair used the relocated planes: {words}"
    s.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_near_singular_corner_through_two_stage_equals_plain_sequence(eng, seed):
    """A corner with almost no texture makes pixel 0's determinant tiny and its flow arbitrary -- huge, negative, Inf or NaN --
    so the shifts of the finer levels land anywhere.  Whatever they are, the two-stage stream pipeline must reproduce the plain
    sequence bit for bit with status word 0 (noise frames whose top-left region is flat up to +-1 LSB, several seeds)."""
    import torch

    w, h, L, win = 1280, 768, 5, 9
    rng = np.random.default_rng(seed)
    host = []
    for i in range(7):
        f = rng.integers(0, 256, (h, w), dtype=np.uint8)
        flat = rng.integers(40, 200)
        n = 48 + 16 * (i % 3)
        f[:n, :n] = flat + rng.integers(-1, 2, (n, n))
        if i % 2:
            f[: n // 2, : n // 2] = flat   # exactly flat: singular (NaN / Inf flows, no shift)
        host.append(f)
    frames = [torch.from_numpy(f).cuda() for f in host]
    want = _plain_sequence(eng, frames, w, h, L, win)
    for batch in (1, 4):
        s = eng.Session(w, h, L, win, "lk_float", stream_batch=batch, borrow_frames=True, two_stage=True)
        got = _stream_all_pairs(s, frames, L, batch)
        for p in got:
            for k in range(L):
                assert_same(got[p][k], want[p][k], f"seed {seed} batch {batch} pair {p} level {k}")
        assert s.corner_status() == 0
        s.close()


@pytest.mark.parametrize("batch,borrow", [(2, False), (4, False), (8, False), (16, False), (1, True), (4, True), (8, True), (16, True),
                                          (1, "two_stage"), (2, "two_stage"), (4, "two_stage"), (8, "two_stage"), (16, "two_stage"),
                                          (3, False), (5, True), (5, "two_stage"), (10, "two_stage")])   # (any B up to 16 is accepted)
@pytest.mark.parametrize("cfg", [(1280, 720, 4, 9, "lk_float", 1, 12), (640, 480, 3, 5, "compat_cpu", 1, 9), (1920, 1088, 5, 7, "lk_float", 4, 11),
                                 (640, 480, 6, 9, "lk_float", 1, 14), (250, 186, 2, 7, "lk_float", 1, 7), (1280, 768, 7, 5, "lk_float", 2, 9),
                                 (320, 240, 3, 7, "lk_float", 2, 53)])  # the last: more than three full ticks of sixteen frames
def test_multi_frame_stream_ticks_equal_plain_sequence(eng, cfg, batch, borrow):
    """ofx_params.stream_batch = B: one launch per B frames (the LK items of B pairs share a launch: taller strips, 1/B of
    the launches).  Only every B-th call launches; pairs complete B at a time and are read through ofx_session_flow_of.
    Every pair must carry the bits of the plain sequence -- also for row-sharded sessions with local corner flows (third
    config: 4 logical ranks) and for frame counts that are not a multiple of B (the tail goes out when the stream is
    drained).  borrow: ofx_params.borrow_frames -- no level-0 copy, the LK and corner stages read the frame buffers
    (padded, dirty padding) in place; "two_stage": borrowed frames and ofx_params.stream_two_stage -- the corner blocks build
    the patch pyramids they read, a pair is complete one tick earlier."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win, mode, R, nf = cfg
    B = batch
    two_stage = borrow == "two_stage"
    borrow = bool(borrow)
    if B * L > 80:
        pytest.skip("stream_batch * levels exceeds OFX_MAX_LK_ITEMS")
    pitch = (w + 3) // 4 * 4 + 8
    def padded(a):
        buf = torch.full((h, pitch), 0x5A, dtype=torch.uint8, device="cuda")
        buf[:, :w] = torch.from_numpy(a).cuda()
        return buf[:, :w]
    frames = [padded(synth.smooth_pair(w, h, 1.2 * i, -0.6 * i, seed=41)[1]) for i in range(nf)]
    plain = eng.Session(w, h, L, win, mode)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want[i] = [plain.flow_host(k) for k in range(L)]
        plain.swap()
    plain.close()

    if R == 1:
        ranks = [eng.Session(w, h, L, win, mode, stream_batch=B, borrow_frames=borrow, two_stage=two_stage)]
    else:
        ranks = [eng.Session(w, h, L, win, mode, shard=ShardPlan(w, h, L, win, r, R), local_corner=True, stream_batch=B,
                             borrow_frames=borrow, two_stage=two_stage) for r in range(R)]
    got = {}
    for s in ranks:
        s.stream_begin()
    seen = 0
    def snap(done):
        nonlocal seen
        if done >= 1:
            assert done - seen <= B
            for p in range(max(seen + 1, done - B + 1), done + 1):   # the newest B pairs are readable
                got[p] = [[s.flow_of(p, k)[0].clone() for k in range(L)] for s in ranks]
            seen = done
    for i in range(nf):
        dones = [s.stream_submit(frames[i]) for s in ranks]
        assert len(set(dones)) == 1
        if i % B != B - 1:
            assert dones[0] == -1   # the frame is only remembered
        snap(dones[0])
    while True:
        dones = [s.stream_drain() for s in ranks]
        assert len(set(dones)) == 1
        if dones[0] == -2:
            break
        snap(dones[0])
    torch.cuda.synchronize()
    assert sorted(got) == list(range(1, nf))
    for p in range(1, nf):
        for k in range(L):
            full = torch.cat([got[p][r][k] for r in range(len(ranks))], dim=0).cpu().numpy()
            assert_same(full, want[p][k], f"{mode} pair {p} level {k}")
    for s in ranks:
        s.close()


def test_stream_pipeline_fuzz(eng):
    """Thirty seeded random configurations of the stream pipeline (size, levels, window, mode, frames per tick, borrowed
    frames, row sharding with local corner flows, padded frame buffers) against the plain sequence: tools/fuzz_stream.py
    (run it with a larger count and other seeds for a longer soak; in round 1 some 4 000 configurations of six seeds passed
    on the final kernels -- one of them a sharded pair whose shift left the shard's halo, reported as such by the status
    word -- and 1 700 of tools/fuzz_plain_vs_oracle.py against the oracl; at the end of round 2, with the folded priming, the sliding box sums of every
    radius, sixteen frames per launch and the two-stage pipeline in the draw: 3 400 and 2 100, no failure)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_stream", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_stream.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(30, 3, verbose=False) == 0


def test_sharded_driver_single_rank_pipelined(eng):
    """parallel.ShardedFlow with world = 1 drives the staged halves (stage_frame / corner_flows / stage_shift /
    solve_staged on the session's aux stream) exactly as bench.py --gpus N does on every rank; the broadcast is the only
    step a single rank skips."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardedFlow

    w, h, L, win = 640, 480, 3, 7
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.1 * i, 0.6 * i, seed=9)[1]).cuda() for i in range(5)]
    plain = eng.Session(w, h, L, win, "lk_float")
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    drv = ShardedFlow(w, h, L, win, "lk_float", 0, 1)
    drv.push_frame(frames[0])
    for i in range(1, 5):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        drv.step(frames[i])
        torch.cuda.synchronize()
        for k in range(L):
            assert_same(drv.gather_flow(k).cpu().numpy(), plain.flow_host(k), f"pair {i} level {k}")
        plain.swap()
    plain.close()
    drv.session.close()


# ---- extension: refinement iterations (pinned by the oracle restatement only) ----------------------------

def test_warp_bilinear_matches_oracle(eng, oracle):
    rng = np.random.default_rng(3)
    for (w, h) in ((64, 48), (301, 37), (2, 2), (1, 5)):
        img = synth.random_pair(w, h, seed=w)[0]
        flow = (rng.normal(size=(h, w, 2)) * 6).astype(np.float32)
        flow[0, 0] = (np.nan, 1.0)
        flow[h - 1, w - 1] = (1e30, -1e30)
        flow[h // 2, w // 2] = (np.inf, 0.0)
        assert_same(eng.warp_u8(img, flow, float(oracle.ITER_SCALE)), oracle.warp_bilinear_u8(img, flow), f"warp {w}x{h}")


@pytest.mark.parametrize("cfg", [(160, 120, 1, 9, 3), (320, 240, 3, 7, 5), (256, 192, 4, 5, 2)])
def test_iterative_refinement_matches_oracle(eng, oracle, cfg):
    """iters > 1: iteration 1 is the reference level, each further one warps the shifted next image by the flow so far
    (bilinear, u8) and accumulates; the shift vectors between levels stay the reference's.  Bit-exact vs the restatement."""
    w, h, L, win, iters = cfg
    p, n = synth.smooth_pair(w, h, 1.2, -0.8)
    got = eng.flow_pair(p, n, L, win, "lk_float", iters=iters)
    want = oracle.flow_pair_iter(p, n, L, win, iters)
    for k in range(L):
        assert_same(got[k], want[k], f"iters={iters} L{k}")
    one = eng.flow_pair(p, n, L, win, "lk_float", iters=1)
    ref, _, _ = oracle.flow_pair(synth.to_3ch(p), synth.to_3ch(n), L, win, "lk_float", exact_sums=True)
    for k in range(L):
        assert_same(one[k], ref[k], f"iters=1 is the reference, L{k}")


@pytest.mark.parametrize("mode", ["lk_float", "lk_float_fast"])
@pytest.mark.parametrize("cfg", [(324, 204, 3, 9, 4), (640, 360, 2, 15, 3), (250, 130, 2, 5, 6), (517, 259, 1, 23, 3)])
def test_fused_warp_of_the_next_iteration_equals_the_warp_launch(eng, monkeypatch, cfg, mode):
    """From the second refinement iteration on, the accumulating launch of iteration j also writes the warped image iteration
    j + 1 reads (csrc/lk_body_warp.h: per pixel two dwords of taps through a buffer resource, no general form), so only the first
    refinement iteration runs ofx_warp_levels.  OFX_ITER_FUSED=0 keeps one warp launch per iteration: both must give the same
    bits -- on frames that drive the warp everywhere it can go: flat blocks (the reference's unguarded solve leaves NaN / Inf
    there: "no warp"), noise (huge finite flows: taps clamped to all four borders, neighbours far apart), odd widths (a row's
    last dword, the ragged end of a tile), every window class of the box sums."""
    w, h, L, win, iters = cfg
    p, n = synth.random_pair(w, h, seed=w + win)
    sp, sn = synth.smooth_pair(w, h, 1.5, -0.9, seed=w)
    p[: h // 2], n[: h // 2] = sp[: h // 2], sn[: h // 2]          # smooth texture above, noise below
    p[h // 3: h // 3 + 40, w // 4: w // 4 + 90] = 77                # a flat block in both frames: 0 / 0 in its windows
    n[h // 3: h // 3 + 40, w // 4: w // 4 + 90] = 77
    monkeypatch.setenv("OFX_ITER_FUSED", "0")
    want = eng.flow_pair(p, n, L, win, mode, iters=iters)
    monkeypatch.setenv("OFX_ITER_FUSED", "1")
    got = eng.flow_pair(p, n, L, win, mode, iters=iters)
    assert not np.isfinite(want[0]).all(), "the frames were meant to produce non-finite flows"
    for k in range(L):
        assert_same(got[k], want[k], f"{mode} {w}x{h} win {win} iters {iters}: level {k}")


def test_iterative_refinement_converges_to_the_translation(eng):
    """Known answer: a smooth texture translated by (0.6,-0.4) px.  Flow is in the reference's units (15/8 of a pixel,
    SURVEY 8a row 10); refinement must move the median estimate closer to the truth than the single reference pass."""
    p, n = synth.smooth_pair(640, 480, 0.6, -0.4)
    truth = np.array([0.6, -0.4])
    err = []
    for iters in (1, 4):
        fl = eng.flow_pair(p, n, 1, 9, "lk_float", iters=iters)[0]
        err.append(np.abs(np.nanmedian(fl.reshape(-1, 2), axis=0) * 8.0 / 15.0 - truth).max())
    assert err[1] < err[0] and err[1] < 0.02, err


def test_session_rejects_bad_configs(eng):
    from cuda_optical_flow_2_amd.lib import OfxError

    with pytest.raises(OfxError):
        eng.Session(100, 50, 3, 9)       # 50>>1 = 25 is odd but gets downsampled again
    with pytest.raises(OfxError):
        eng.Session(64, 48, 3, 8)        # even window
    s = eng.Session(64, 48, 2, 5)
    with pytest.raises(OfxError):
        s.run_flow()                     # no frames yet
    s.close()


# ---- the gpu:: drop-in surface (mangled C++ symbols, host pointers) -------------------------------------------------------

def test_gpu_namespace_primitives_golden(gpu, golden):
    g = golden("primitives")
    img, gray = g["img"], g["gray"]
    assert_same(gpu.grayscale_avg(img), gray, "gpu::grayscale_avg")
    for nm, m in (("dx", gpu.Dx_3x3), ("dy", gpu.Dy_3x3), ("dt", gpu.Dt_3x3), ("gaus", gpu.GAUS_KERNEL_3x3)):
        for variant in ("conv_3ch_1ch_constant", "conv_3ch_1ch_tiled"):
            assert_same(gpu.conv_3ch_1ch(gray, m, variant=variant), g["conv1_" + nm], f"gpu::{variant} {nm}")
        for variant in ("conv_3ch_2d", "conv_3ch_2d_constant"):
            assert_same(gpu.conv_3ch(img, m, 3, 3, variant), g["conv3_" + nm], f"gpu::{variant} {nm}")
    assert_same(gpu.conv_3ch_1ch(gray, g["mask5"], 5, 5), g["conv1_m5"], "gpu::conv_3ch_1ch_constant 5x5")
    assert_same(gpu.conv_3ch(img, g["mask5"], 5, 5), g["conv3_m5"], "gpu::conv_3ch_2d 5x5")
    assert_same(gpu.gauss_pyramid(img, 2)[1], g["down"], "gpu::gauss_pyramid")
    for ww, wh in ((3, 3), (5, 5), (7, 7), (9, 9), (15, 15), (19, 19), (5, 9), (4, 6)):
        for variant in ("srm_1ch", "srm_1ch_tiled"):
            assert_same(gpu.srm_1ch(g["a"], g["b"], ww, wh, variant), g[f"srm_{ww}x{wh}"], f"gpu::{variant} {ww}x{wh}")


@pytest.mark.parametrize("shape", [(41, 33), (259, 70), (520, 97), (1000, 41), (8, 9), (255, 64), (257, 40)])
def test_window_sums_on_the_march_equal_the_oracle(gpu, oracle, shape):
    """gpu::srm_1ch / gpu::srm_1ch_float run as marches since round 4 (csrc/srm_march.hip: sliding integer windows through a wave's
    LDS row; float products formed once and added in the reference's row-major tap order): every window shape the entry points
    take -- odd, even, wider than high, a window larger than the image -- on widths around the 256-column tile and the 248 / 250
    output columns of a tile, odd widths (unaligned rows, a ragged last store), uniform-random bytes (sums far beyond 2^24) and
    floats with NaN / Inf next to the border: int32 bit for bit, float bit for bit against the restatement of OptFlowGpu.cu:1549-1588."""
    w, h = shape
    rng = np.random.default_rng(w * 1000 + h)
    a = rng.integers(0, 256, (h, w), dtype=np.uint8)
    b = rng.integers(0, 256, (h, w), dtype=np.uint8)
    fa = (rng.normal(size=(h, w)) * 300).astype(np.float32)
    fb = (rng.normal(size=(h, w)) * 300).astype(np.float32)
    fa[0, 0], fa[h - 1, w - 1], fb[h // 2, 0], fa[h // 2, w - 1] = np.nan, np.inf, -np.inf, -0.0
    for ww, wh in ((9, 9), (19, 19), (3, 3), (5, 9), (4, 6), (23, 7), (2, 2), (1, 1), (15, 15), (31, 3), (7, 25)):
        assert_same(gpu.srm_1ch(a, b, ww, wh), oracle.srm_1ch(a, b, ww, wh), f"gpu::srm_1ch {w}x{h} {ww}x{wh}")
    for ww, wh in ((19, 19), (9, 9), (3, 7), (4, 6), (1, 1), (23, 5), (10, 3)):
        assert_same(gpu.srm_1ch_float(fa, fb, ww, wh), oracle.srm_1ch_f32(fa, fb, ww, wh), f"gpu::srm_1ch_float {w}x{h} {ww}x{wh}")


@pytest.mark.parametrize("shape", [(16, 9), (17, 5), (255, 33), (1000, 21), (260, 4)])
def test_channel0_correlations_four_pixels_per_thread_equal_the_oracle(gpu, oracle, shape):
    """gpu::conv_3ch_1ch_constant / _tiled (int accumulator truncated after every tap, OptFlowGpu.cu:380-425 == OptFlowCPU.cpp:75-109)
    and gpu::conv_3ch_1ch_tiled_uchar_float (float accumulator, :1040-1090) since round 4 give a thread four adjacent pixels and
    fetch their channel-0 bytes as dwords (csrc/primitives.hip, conv_3ch_1ch_x4_kernel); the threads at the image's edges keep the
    pixel-by-pixel loop.  Every mask of the reference's tables plus fractional, negative, 2- and 4-wide ones, on widths that end
    inside a thread's four pixels, uniform-random bytes: bit for bit."""
    w, h = shape
    rng = np.random.default_rng(w * 131 + h)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    masks = [(gpu.Dx_3x3, 3, 3), (gpu.Dy_3x3, 3, 3), (gpu.Dt_3x3, 3, 3), (gpu.GAUS_KERNEL_3x3, 3, 3), (gpu.GAUS_KERNEL_5x5, 5, 5), (gpu.Dx_5x5, 5, 5),
             (np.array([0.3, -1.7, 2.2, 0.0, -0.45, 1.0], np.float32), 2, 3), (rng.normal(size=12).astype(np.float32) * 3, 4, 3),
             (rng.normal(size=15).astype(np.float32), 5, 3), (rng.normal(size=27).astype(np.float32), 3, 9)]
    for m, mw, mh in masks:
        m = np.asarray(m, np.float32)
        assert_same(gpu.conv_3ch_1ch(img, m, mw, mh), oracle.conv_3ch_to_1ch(img, m, mw, mh), f"conv u8 {w}x{h} mask {mw}x{mh}")
        assert_same(gpu.conv_3ch_1ch_float(img, m, mw, mh), oracle.conv_3ch_to_1ch_f32(img, m, mw, mh), f"conv f32 {w}x{h} mask {mw}x{mh}")


def test_gpu_namespace_float_primitives(gpu, oracle):
    rng = np.random.default_rng(11)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for m in (gpu.Dx_3x3, gpu.Dy_3x3, gpu.Dt_3x3, gpu.Dt_3x3_n, gpu.GAUS_KERNEL_5x5):
        k = 5 if m.size == 25 else 3
        assert_same(gpu.conv_3ch_1ch_float(img, m, k, k), oracle.conv_3ch_to_1ch_f32(img, m, k, k), "conv f32")
    a = rng.normal(size=(33, 41)).astype(np.float32) * 100
    b = rng.normal(size=(33, 41)).astype(np.float32) * 100
    for ww, wh in ((19, 19), (9, 9), (3, 7), (4, 6)):
        assert_same(gpu.srm_1ch_float(a, b, ww, wh), oracle.srm_1ch_f32(a, b, ww, wh), f"gpu::srm_1ch_float {ww}x{wh}")
    s = [rng.integers(-5000, 5000, (20, 30)).astype(np.int32) for _ in range(5)]
    s[0], s[1] = np.abs(s[0]), np.abs(s[1])
    for k in range(3):
        s[k][0, 0] = 0
    assert_flow_close(gpu.inverse_matrix(*s), oracle.inverse_matrix_i32(*s), 1, "gpu::inverse_matrix")
    sf = [x.astype(np.float32) * 1.5 for x in s]
    assert_flow_close(gpu.inverse_matrix_float(*sf), oracle.inverse_matrix_f32(*sf), 1, "gpu::inverse_matrix_float")
    assert_same(gpu.conv_3ch(img, gpu.Dt_3x3_n, 3, 3, "conv_3ch_tiled"),
                np.stack([oracle.conv_3ch_to_1ch_f32(np.repeat(img[:, :, c:c + 1], 3, 2), gpu.Dt_3x3_n).astype(np.int32).astype(np.uint8)
                          for c in range(3)], axis=2), "gpu::conv_3ch_tiled (float accumulators)")


def test_gpu_calc_opt_flow_matches_reference_composition(gpu, oracle):
    """gpu::calc_opt_flow (window 19, Dt_3x3) over a 3-level pyramid, host pointers, vs the oracle's restatement of
    OptFlowGpu.cu:1909-1979."""
    p, n = synth.smooth_pair(160, 120)
    got, gp, gn = gpu.flow_pair(synth.to_3ch(p), synth.to_3ch(n), 3)
    want, pp, npyr = oracle.flow_pair(synth.to_3ch(p), synth.to_3ch(n), 3, 19, "lk_float", exact_sums=True)
    for k in range(3):
        assert_same(gn[k], npyr[k], f"pyramid L{k}")
        assert_same(got[k], want[k], f"flow L{k}")
    # and against the float-order variant within the solve tolerance on well-conditioned pixels
    loose, _, _ = oracle.flow_pair(synth.to_3ch(p), synth.to_3ch(n), 3, 19, "lk_float", exact_sums=False)
    ok = np.isfinite(loose[0]) & np.isfinite(got[0])
    assert np.abs(got[0][ok] - loose[0][ok]).max() < 1e-2 * (1 + np.abs(loose[0][ok]).max())


def test_gpu_bilinear_filter_golden(gpu, golden):
    g = golden("bilateral")
    assert_same(gpu.bilinear_filter(g["gray"], g["gray"], 9, 9, 2.0, 10.0), g["out_gray_9"], "gpu::bilinear_filter 9x9 (main.cu:240)")
    assert_same(gpu.bilinear_filter(g["img"], g["gray"], 5, 5, 1.5, 20.0), g["out_color_5"], "gpu::bilinear_filter 5x5")


# ---- BASELINE full sizes: size-independent properties -----------------------------------------------------------------

FULL = [(1920, 1080, 4, 7), (3840, 2160, 5, 9), (7680, 4320, 6, 15)]  # BASELINE configs 2, 3 and 5 (the maximum size)


@pytest.mark.parametrize("cfg", FULL)
def test_full_size_crops_match_oracle(eng, oracle, cfg):
    """A level-0 interior block depends only on its (radius+1)-pixel neighbourhood, so oracle crops pin the 4K/1080p
    result without running the oracle on the whole frame."""
    w, h, levels, win = cfg
    p, n = synth.smooth_pair(w, h)
    r = win // 2 + 1
    whole = eng.lk_level(p, n, win, "lk_float")
    rng = np.random.default_rng(w)
    boxes = [(0, 0), (w - 160, h - 96), (w - 160, 0), (0, h - 96)] + [(int(rng.integers(0, w - 160)), int(rng.integers(0, h - 96))) for _ in range(4)]
    for (x0, y0) in boxes:
        x1, y1 = x0 + 160, y0 + 96
        cx0, cy0, cx1, cy1 = max(0, x0 - r), max(0, y0 - r), min(w, x1 + r), min(h, y1 + r)
        # keep true image borders as borders, cut everywhere else with a full halo
        want = _oracle_level(oracle, p[cy0:cy1, cx0:cx1], n[cy0:cy1, cx0:cx1], win, "lk_float")
        assert_same(whole[y0:y1, x0:x1], want[y0 - cy0:y1 - cy0, x0 - cx0:x1 - cx0], f"crop at ({x0},{y0})")


@pytest.mark.parametrize("cfg", FULL)
def test_full_size_properties(eng, cfg):
    import torch

    w, h, levels, win = cfg
    p, n = synth.smooth_pair(w, h)
    # (1) identical frames: every temporal sum is 0, so the flow is exactly (+-0, +-0) wherever the system is regular
    same = eng.lk_level(p, p, win, "lk_float")
    fin = np.isfinite(same)
    assert fin.mean() > 0.99 and (same[fin] == 0).all()
    # (2) two row blocks with halo == whole frame, bit for bit
    whole = eng.lk_level(p, n, win, "lk_float")
    cut, halo = h // 2 + 3, win // 2 + 1
    top = eng.lk_level(p, n, win, "lk_float", rows=(0, cut), buf_rows=(0, cut + halo))
    bot = eng.lk_level(p, n, win, "lk_float", rows=(cut, h), buf_rows=(cut - halo, h))
    assert_same(np.concatenate([top, bot]), whole, "two-block sharding")
    # (3) the session pipeline reproduces the stand-alone level at the top of the pyramid and is deterministic
    s = eng.Session(w, h, levels, win, "lk_float")
    s.push_frame_host(p)
    s.set_frame_host(n)
    s.build_pyramid()
    s.run_flow()
    torch.cuda.synchronize()
    a = [s.flow_host(k) for k in range(levels)]
    s.run_flow()
    torch.cuda.synchronize()
    b = [s.flow_host(k) for k in range(levels)]
    for k in range(levels):
        assert_same(a[k], b[k], f"determinism L{k}")
    top_prev, g = s.plane(0, levels - 1)
    top_next, _ = s.plane(1, levels - 1)
    lone = eng.lk_level(top_prev[:, : g.w].cpu().numpy(), top_next[:, : g.w].cpu().numpy(), win, "lk_float")
    assert_same(a[levels - 1], lone, "top level")
    # (4) median level-0 residual on the smooth texture is finite
    med = np.nanmedian(a[0].reshape(-1, 2), axis=0)
    assert np.isfinite(med).all()
    s.close()


# ---- solve options: OFX_MODE_LK_FLOAT_FAST (<= 1 ulp) and the determinant guard ------------------------------------------

def _stripes(w, h, ang, shift):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    t = np.cos(ang) * xx + np.sin(ang) * yy
    return np.clip(127 + 100 * np.sin((t - shift) * 0.3), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("case", ["smooth", "random", "stripes45", "stripes_x", "stripes17", "flat", "identical"])
def test_fast_solve_is_within_one_ulp_of_the_replay(eng, case):
    """OFX_MODE_LK_FLOAT_FAST: numerators first, reciprocal to 2^-44 (csrc/lk_solve.h).  SURVEY 8c's tolerance for the solve:
    identical NaN / Inf positions, finite values within 1 float ulp of the replayed reference solve -- on textures,
    noise, and on the degenerate inputs where the 2x2 system is singular (exactly: axis-parallel and 45-degree stripes,
    flat frames; nearly: 17-degree stripes), where every pixel with det == 0 must take the replay's path."""
    w, h, win = 1000, 600, 9
    if case == "smooth":
        p, n = synth.smooth_pair(w, h, 2.0, 1.0)
    elif case == "random":
        p, n = synth.random_pair(w, h)
    elif case == "flat":
        p = np.full((h, w), 77, np.uint8); n = np.full((h, w), 80, np.uint8)
    elif case == "identical":
        p, _ = synth.smooth_pair(w, h); n = p.copy()
    else:
        ang = {"stripes45": np.pi / 4, "stripes_x": 0.0, "stripes17": 0.3}[case]
        p, n = _stripes(w, h, ang, 0.0), _stripes(w, h, ang, 0.7)
    exact = eng.lk_level(p, n, win, "lk_float")
    fast = eng.lk_level(p, n, win, "lk_float_fast")
    assert_flow_close(fast, exact, 1, f"fast solve, {case}")
    if case in ("smooth", "random"):
        same = (fast == exact) | (np.isnan(fast) & np.isnan(exact))
        assert same.mean() > 0.9999, same.mean()   # the two formulations agree to the bit almost everywhere


def test_fast_solve_through_every_path(eng):
    """lk_float_fast through the session: plain sequence, stream pipeline (two frames per launch) and 3 row shards give the
    same bits as each other (the choice of formula is per pixel, never per wave), and stay within 1 ulp of lk_float."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win, nf = 640, 480, 3, 9, 6
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.3 * i, -0.7 * i, seed=11)[1]).cuda() for i in range(nf)]

    def plain(mode):
        s = eng.Session(w, h, L, win, mode)
        s.set_frame_device(frames[0]); s.build_pyramid(); s.swap()
        out = {}
        for i in range(1, nf):
            s.set_frame_device(frames[i]); s.build_pyramid(); s.run_flow()
            torch.cuda.synchronize()
            out[i] = [s.flow_host(k) for k in range(L)]
            s.swap()
        s.close()
        return out

    exact, fast = plain("lk_float"), plain("lk_float_fast")
    for i in exact:
        for k in range(L):
            assert_flow_close(fast[i][k], exact[i][k], 1, f"pair {i} level {k}")
    for R in (1, 3):
        ranks = [eng.Session(w, h, L, win, "lk_float_fast", stream_batch=2, shard=None if R == 1 else ShardPlan(w, h, L, win, r, R),
                             local_corner=R > 1) for r in range(R)]
        got, seen = {}, 0
        for s in ranks:
            s.stream_begin()

        def snap(done):
            nonlocal seen
            if done >= 1:
                for p in range(max(seen + 1, done - 1), done + 1):
                    got[p] = [torch.cat([s.flow_of(p, k)[0] for s in ranks], dim=0).cpu().numpy() for k in range(L)]
                seen = done
        for f in frames:
            snap([s.stream_submit(f) for s in ranks][0])
        while True:
            d = [s.stream_drain() for s in ranks][0]
            if d == -2:
                break
            snap(d)
        for i in range(1, nf):
            for k in range(L):
                assert_same(got[i][k], fast[i][k], f"{R} rank(s), stream: pair {i} level {k}")
        for s in ranks:
            s.close()


def test_determinant_guard(eng, oracle):
    """ofx_params.min_det (extension): pixels whose determinant, rounded to float, is below the threshold get the
    flow (0, 0) instead of the reference's unguarded quotient; everything else keeps the reference's bits; min_det = 0 is
    the reference.  Checked on a single level against the oracle's sums (the guard is a function of Sxx, Syy, Sxy only), in
    both modes, and through a whole session (where the guarded pixel 0 also feeds the shift vectors) for consistency
    between the plain and the stream path."""
    import torch

    w, h, win = 640, 360, 9
    p, n = synth.smooth_pair(w, h, 1.0, 0.5)
    p[100:200, 200:400] = 90; n[100:200, 200:400] = 93          # a flat patch: det == 0 inside it
    for mode in ("lk_float", "compat_cpu"):
        ref = eng.lk_level(p, n, win, mode)
        sums = eng.lk_level(p, n, win, mode, want_sums=True).astype(np.float64)
        if mode == "lk_float":
            a, d, b = (sums[i].astype(np.float32).astype(np.float64) for i in (0, 1, 2))
        else:
            a, d, b = sums[0], sums[1], sums[2]
        det = (a * d - b * b).astype(np.float32)
        thr = float(np.percentile(det, 30))
        s = eng.Session(w, h, 1, win, mode, min_det=thr)
        s.push_frame_host(p); s.set_frame_host(n); s.build_pyramid(); s.run_flow()
        torch.cuda.synchronize()
        got = s.flow_host(0)
        s.close()
        low = ~(det >= np.float32(thr))
        assert low.sum() > 1000 and (~low).sum() > 1000
        assert (got[low] == 0).all(), "guarded pixels must be exactly (0, 0)"
        assert_same(got[~low], ref[~low], f"{mode}: pixels above the threshold keep the reference's bits")
        assert np.isnan(ref[150, 300]).all() and (got[150, 300] == 0).all()   # the flat patch: NaN in the reference
    # whole sessions: plain == stream with the guard on, 3 levels
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.1 * i, 0.4 * i, seed=3)[1]).cuda() for i in range(5)]
    plain = eng.Session(w, h, 3, win, "lk_float", min_det=5e9)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, 5):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want[i] = [plain.flow_host(k) for k in range(3)]
        plain.swap()
    plain.close()
    assert any((want[i][0] == 0).all(axis=2).mean() > 0.01 for i in want)
    st = eng.Session(w, h, 3, win, "lk_float", min_det=5e9)
    st.stream_begin()
    got = {}
    for f in frames:
        d = st.stream_submit(f)
        if d >= 1:
            got[d] = [st.flow(k)[0].clone() for k in range(3)]
    while True:
        d = st.stream_drain()
        if d == -2:
            break
        if d >= 1:
            got[d] = [st.flow(k)[0].clone() for k in range(3)]
    torch.cuda.synchronize()
    for i in want:
        for k in range(3):
            assert_same(got[i][k].cpu().numpy(), want[i][k], f"guard on: stream pair {i} level {k}")
    st.close()


@pytest.mark.parametrize("borrow", [False, True, "two_stage"])
@pytest.mark.parametrize("cfg", [(640, 480, 3, 9, 3, 1), (640, 480, 4, 7, 5, 2), (1280, 720, 4, 9, 2, 4), (256, 192, 2, 5, 4, 8)])
def test_streamed_refinement_iterations_equal_the_pair_at_a_time_path(eng, cfg, borrow):
    """iters > 1 through the stream pipeline: the tick's LK stage is iteration 1 of its B pairs, every further iteration is
    ONE warp launch and ONE accumulating LK launch over all levels of all B pairs.  Same arithmetic as ofx_session_run_flow
    with iters > 1 (itself bit-exact against the oracle's orc_lk_iter_level): every pair, every level, bit for bit; also for a
    frame count that leaves a partial tick to the drain."""
    import torch

    w, h, L, win, iters, B = cfg
    nf = 2 * B + 3
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.2 * i, -0.7 * i, seed=19)[1]).cuda() for i in range(nf)]
    plain = eng.Session(w, h, L, win, "lk_float", iters=iters)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want[i] = [plain.flow_host(k) for k in range(L)]
        plain.swap()
    plain.close()
    s = eng.Session(w, h, L, win, "lk_float", iters=iters, stream_batch=B, borrow_frames=bool(borrow), two_stage=borrow == "two_stage")
    s.stream_begin()
    got, seen = {}, 0

    def snap(done):
        nonlocal seen
        if done >= 1:
            for p in range(max(seen + 1, done - B + 1), done + 1):
                got[p] = [s.flow_of(p, k)[0].clone() for k in range(L)]
            seen = done
    for f in frames:   # (every frame its own buffer, pitch = width = a multiple of 64: what borrowed frames need here)
        snap(s.stream_submit(f))
    while True:
        d = s.stream_drain()
        if d == -2:
            break
        snap(d)
    torch.cuda.synchronize()
    assert sorted(got) == list(range(1, nf))
    for p in range(1, nf):
        for k in range(L):
            assert_same(got[p][k].cpu().numpy(), want[p][k], f"iters={iters} B={B}: pair {p} level {k}")
    s.close()
    if borrow:
        from cuda_optical_flow_2_amd.lib import OfxError
        bad = eng.Session(w, h, L, win, "lk_float", iters=iters, stream_batch=1, borrow_frames=True)
        bad.stream_begin()
        padded = [torch.zeros((h, w + 64), dtype=torch.uint8, device="cuda")[:, :w] for _ in range(4)]
        with pytest.raises(OfxError):   # the launches of an iteration address a level's planes with one pitch: the session's
            for f in padded:
                bad.stream_submit(f)
        bad.close()


@pytest.mark.parametrize("cfg", [(640, 480, 3, 9, 3, 3, 2, 8), (1280, 768, 4, 7, 4, 4, 1, 8), (1920, 1088, 5, 5, 2, 8, 4, 48)])
def test_sharded_streamed_iterations_equal_the_unsharded_path(eng, cfg):
    """iters > 1 on row-sharded sessions (stream pipeline, local corner flows, nothing passed between the ranks): iteration j
    is computed on (radius + 1) * (iters - j) rows beyond a rank's block, so that the warp of iteration j + 1 finds the flow
    of every row its LK stencils touch in the rank's own buffers (parallel.ShardPlan(iters=...) sizes the halo).  R logical
    ranks on one device, put together == the unsharded pair-at-a-time path, bit for bit; status words stay 0.  The warp
    follows the flow, so a rank also holds `warp_margin` rows of slack: a 5x5 window on this texture produces outliers of
    several rows (third configuration: 48 rows of slack; with the default 8 the same run differs in a few pixels next to a
    cut AND says so: bit 16 + level of the status word)."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win, iters, R, B, wm = cfg
    nf = 2 * B + 2
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.1 * i, 0.6 * i, seed=29)[1]).cuda() for i in range(nf)]
    plain = eng.Session(w, h, L, win, "lk_float", iters=iters)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want[i] = [plain.flow_host(k) for k in range(L)]
        plain.swap()
    plain.close()
    def run(warp_margin, strict=True):
        ranks = [eng.Session(w, h, L, win, "lk_float", shard=ShardPlan(w, h, L, win, r, R, iters=iters, warp_margin=warp_margin),
                             local_corner=True, stream_batch=B, iters=iters, strict=strict) for r in range(R)]
        return ranks

    ranks = run(wm)
    got, seen = {}, 0
    for s in ranks:
        s.stream_begin()

    def snap(done):
        nonlocal seen
        if done >= 1:
            for p in range(max(seen + 1, done - B + 1), done + 1):
                got[p] = [torch.cat([s.flow_of(p, k)[0] for s in ranks], dim=0).cpu().numpy() for k in range(L)]
            seen = done
    for f in frames:
        snap([s.stream_submit(f) for s in ranks][0])
    while True:
        d = [s.stream_drain() for s in ranks][0]
        if d == -2:
            break
        snap(d)
    assert sorted(got) == list(range(1, nf))
    for p in range(1, nf):
        for k in range(L):
            assert_same(got[p][k], want[p][k], f"{R} ranks, iters={iters}: pair {p} level {k}")
    for s in ranks:
        assert s.corner_status() == 0
        s.close()
    if wm > 8:   # the same with the default slack: whatever differs is flagged
        ranks = run(8, strict=False)
        for s in ranks:
            s.stream_begin()
            for f in frames:
                s.stream_submit(f)
            while s.stream_drain() != -2:
                pass
        torch.cuda.synchronize()
        st = [s.corner_status() for s in ranks]
        for s in ranks:
            s.close()
        assert any(x >> 16 for x in st), [hex(x) for x in st]
    # a plan made without its iterations has too small a halo: the session says so instead of computing something else
    from cuda_optical_flow_2_amd.lib import OfxError
    if (win // 2 + 1) * iters > win // 2 + 1 + 8:   # (the default plan's margin of 8 rows does not cover the iterations' rows)
        with pytest.raises(OfxError):
            eng.Session(w, h, L, win, "lk_float", shard=ShardPlan(w, h, L, win, 1, R), local_corner=True, iters=iters)


_WIDE_SNIPPET = r"""
import sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from cuda_optical_flow_2_amd import engine as eng, synth

def same(a, b):
    return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all().item())

for (w, h, L, win, B) in ((1280, 720, 4, 9, 4), (1000, 564, 3, 7, 2), (640, 480, 3, 15, 2)):
    nf = 3 * B + 2
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.3 * i, -0.7 * i, seed=31)[1]).cuda() for i in range(nf)]
    frames[3] = torch.from_numpy(synth.random_pair(w, h, 7)[0]).cuda()   # (non-finite flows, every clamp)
    s = eng.Session(w, h, L, win, "lk_float", stream_batch=B)
    plain = eng.Session(w, h, L, win, "lk_float")
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    s.stream_begin()
    done = -1
    got = {{}}
    def snap(d):
        for p in range(max(1, d - B + 1), d + 1):
            if p not in got:
                got[p] = [s.flow_of(p, k)[0].clone() for k in range(L)]
    for f in frames:
        d = s.stream_submit(f)
        if d >= 1: snap(d)
    while True:
        d = s.stream_drain()
        if d == -2: break
        if d >= 1: snap(d)
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        for k in range(L):
            assert i in got and same(got[i][k], plain.flow(k)[0]), (w, h, win, i, k)
        plain.swap()
    s.close(); plain.close()
print("wide ok")
"""


def test_march_with_eight_columns_per_lane_is_bit_identical(eng):
    """csrc/lk_body_wide.h (round 4): the LK march with eight columns per lane is selected per process (OFX_LK_COLS=8 for the stream
    tick, OFX_LK_PLAIN_COLS=8 for the pair-at-a-time launch; measured slower, so not the default).  A child process with both set
    streams three configurations (odd width, windows 7 / 9 / 15, a uniform-random frame in the middle) and compares every pair and
    level of the tick with the plain sequence: first the tick on the wide march against the plain launch on the narrow one, then
    the other way round -- the narrow forms are the ones every other test pins to the oracle."""
    import subprocess
    import sys as _sys

    env = dict(os.environ, OFX_LK_COLS="8", OFX_LK_PLAIN_COLS="0")
    code = _WIDE_SNIPPET.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "wide ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
    env = dict(os.environ, OFX_LK_COLS="4", OFX_LK_PLAIN_COLS="8")   # the narrow tick against the wide pair-at-a-time launch
    r = subprocess.run([_sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "wide ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


_PLAIN_FUSED_SNIPPET = r"""
import sys
import numpy as np, torch
sys.path.insert(0, {root!r})
from cuda_optical_flow_2_amd import engine as eng, synth

out = {{}}
for (w, h, L, win, mode) in ((1280, 720, 5, 9, "lk_float"), (640, 480, 4, 15, "compat_cpu"), (400, 304, 3, 7, "lk_float"), (64, 48, 3, 5, "lk_float")):
    frames = [synth.smooth_pair(w, h, 1.3 * i, -0.7 * i, seed=37)[1] for i in range(4)]
    frames[2] = synth.random_pair(w, h, 9)[0]   # (a corner whose vectors are large or not finite: the chain leaves its patch)
    s = eng.Session(w, h, L, win, mode)
    s.set_frame_host(frames[0]); s.build_pyramid(); s.swap()
    for i in range(1, 4):
        s.timing(8)
        s.set_frame_host(frames[i]); s.build_pyramid(); s.run_flow()
        torch.cuda.synchronize()
        out[f"{{w}}_{{mode}}_{{i}}_launches"] = np.array([s.timing_read_kind("pyramid")[2], s.timing_read_kind("corner")[2], s.timing_read_kind("lk")[2]])
        s.timing(0)
        for k in range(L):
            out[f"{{w}}_{{mode}}_{{i}}_{{k}}"] = s.flow_host(k)
        for k in range(L - 1):
            out[f"{{w}}_{{mode}}_{{i}}_uv{{k}}"] = s.uv(k).cpu().numpy().copy()
        s.swap()
    s.close()
np.savez(sys.argv[1], **out)
print("plain ok")
"""


def test_pyramid_launch_that_carries_the_corner_chain_gives_the_same_flows(tmp_path):
    """csrc/pyr_corner.hip (round 4, OFX_PLAIN_FUSED=1; measured slower than the three launches, so not the default): the pair-at-a-time
    path's pyramid launch walks the pair's corner chain in one more block, on a patch pyramid that block builds (the previous
    frame's patch planes are kept from the pair before), and repairs a shift that leaves the patch from the whole level 0.  Child
    processes run three consecutive pairs of four configurations -- the default three launches; the fused two; the fused two with
    the chain's planes narrowed to 48 level-0 pixels so that ordinary frames need the repair too -- and every flow and every shift
    vector must be bit-identical; the fused children must show one pyramid launch, no corner launch, one LK launch per pair."""
    import subprocess
    import sys as _sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for name, extra in (("default", {}), ("fused", {"OFX_PLAIN_FUSED": "1"}), ("fused_narrow", {"OFX_PLAIN_FUSED": "1", "OFX_DEBUG_CORNER_EXTENT": "48"})):
        path = str(tmp_path / f"{name}.npz")
        r = subprocess.run([_sys.executable, "-c", _PLAIN_FUSED_SNIPPET.format(root=root), path], env=dict(os.environ, **extra), capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0 and "plain ok" in r.stdout, (name, r.stdout[-500:], r.stderr[-2000:])
        got[name] = dict(np.load(path))
    for name in ("fused", "fused_narrow"):
        fused_pairs = 0
        for k, v in got["default"].items():
            if k.endswith("_launches"):
                assert v.tolist() == [1, 1, 1], (k, v)
                fused_pairs += int(got[name][k].tolist() == [1, 0, 1])
                continue
            a, b = got[name][k], v
            assert a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all()), (name, k)
        assert fused_pairs >= 9, (name, fused_pairs)   # (the 64 x 48 session's patch is the frame: fused too; at least the three larger ones)
