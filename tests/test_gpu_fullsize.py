"""GPU: the configurations bench.py times, at their own size (BASELINE.json configs 2-4).

The stream pipeline plans differently at every size (number of LK waves per SIMD, strip heights, pyramid block counts),
so the bit-exactness shown at <= 1920x1088 in test_gpu_parity.py does not carry over by itself: these tests run the
benchmarked configuration -- 3840x2160, 5 levels, 9x9, four frames per launch, frames read in place from a ring of 16
padded device buffers -- and compare EVERY pair with the plain pair-at-a-time sequence bit for bit, and whole pairs with
the CPU oracle (it finishes a 4K pair in ~5 s).  The 8-way row-sharded forms of the same pair (BASELINE config 4) run as
eight logical ranks on the one device.
"""
import numpy as np
import pytest

from conftest import assert_same
from cuda_optical_flow_2_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    from cuda_optical_flow_2_amd import engine

    return engine


def _same_bits(a, b) -> bool:
    """device-side equality with NaN == NaN (the flows are compared where they lie: a 4K pair's flow pyramid is 88 MB)"""
    import torch

    return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all().item())


def _plain_sequence(eng, frames, w, h, L, win, mode, iters=1):
    """{pair: [flow level k, device clone]} of set_frame / build_pyramid / run_flow / swap over `frames`"""
    import torch

    plain = eng.Session(w, h, L, win, mode, iters=iters)
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = {}
    for i in range(1, len(frames)):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        want[i] = [plain.flow(k)[0].clone() for k in range(L)]
        plain.swap()
    torch.cuda.synchronize()
    plain.close()
    return want


def _ring(frames_host, w, h, ring_n, pad, fill):
    """ring_n padded device buffers (pitch = w + pad, dirty padding); returns (buffers, views [h, w])"""
    import torch

    bufs = [torch.full((h, w + pad), fill, dtype=torch.uint8, device="cuda") for _ in range(ring_n)]
    return bufs, [b[:, :w] for b in bufs]


def _run_stream(eng, sessions, views, frames_host, B, L, w):
    """Feed len(frames_host) frames through the sessions' stream pipeline; frame i is written into ring slot i % ring
    (in stream order, right before its submit -- a slot is reused a ring's length of submits later, past the 3 * B (2 * B with
    stream_two_stage) the borrow contract asks for).  Returns {pair: [[level k of rank r]]}."""
    import torch

    ring_n = len(views)
    got, seen = {}, 0

    def snap(done):
        nonlocal seen
        if done >= 1:
            for p in range(max(seen + 1, done - B + 1), done + 1):
                got[p] = [[s.flow_of(p, k)[0].clone() for k in range(L)] for s in sessions]
            seen = done

    for s in sessions:
        s.stream_begin()
    for i, f in enumerate(frames_host):
        views[i % ring_n].copy_(torch.from_numpy(f).cuda())
        dones = [s.stream_submit(views[i % ring_n]) for s in sessions]
        assert len(set(dones)) == 1
        snap(dones[0])
    while True:
        dones = [s.stream_drain() for s in sessions]
        assert len(set(dones)) == 1
        if dones[0] == -2:
            break
        snap(dones[0])
    torch.cuda.synchronize()
    return got


# (width, height, levels, window, mode, frames per launch, frames, ofx_params.stream_two_stage)
BENCHED = [
    pytest.param((3840, 2160, 5, 9, "lk_float", 8, 37, True), id="4k-lk_float-batch8-two-stage-the-bench-default"),
    pytest.param((3840, 2160, 5, 9, "lk_float", 4, 22, False), id="4k-lk_float-batch4-three-stage"),
    pytest.param((1920, 1080, 4, 7, "lk_float", 16, 53, True), id="1080p-lk_float-batch16-two-stage"),
    pytest.param((1920, 1080, 4, 7, "lk_float", 8, 27, False), id="1080p-lk_float-batch8"),
    pytest.param((3840, 2160, 5, 9, "compat_cpu", 8, 21, True), id="4k-compat_cpu-batch8-two-stage"),
]


@pytest.mark.parametrize("cfg", BENCHED)
def test_benchmarked_stream_configuration_is_bit_exact(eng, oracle, cfg):
    """bench.py's timed configurations (ofx_params.stream_batch, borrow_frames, stream_two_stage, a ring of padded buffers of
    bench.py's size that is reused while the stream runs) against the plain sequence for every pair and every level, and
    against the oracle for the first and the last pair."""
    import torch

    w, h, L, win, mode, B, nf, two_stage = cfg
    frames = [synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1] for i in range(nf)]   # bench.py's frames: (2,1) px per frame
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    want = _plain_sequence(eng, d_frames, w, h, L, win, mode)
    del d_frames
    ring_n = ((2 if two_stage else 3) * max(B, 4) + 4 + 3) // 4 * 4   # bench.py's ring: 20 buffers for eight frames per launch in two stages
    _, views = _ring(frames, w, h, ring_n, 64, 0xA5)
    s = eng.Session(w, h, L, win, mode, stream_batch=B, borrow_frames=True, two_stage=two_stage)
    got = _run_stream(eng, [s], views, frames, B, L, w)
    s_status = s.corner_status()
    s.close()
    assert sorted(got) == list(range(1, nf))
    for p in range(1, nf):
        for k in range(L):
            assert _same_bits(got[p][0][k], want[p][k]), f"{mode} {w}x{h} B={B}: pair {p} level {k} differs from the plain sequence"
    assert s_status == 0, f"status word {s_status:#x}: a corner shift left its patch"
    for p in (1, nf - 1):
        ref, _, _ = oracle.flow_pair(synth.to_3ch(frames[p - 1]), synth.to_3ch(frames[p]), L, win, mode, exact_sums=True)
        for k in range(L):
            assert_same(got[p][0][k].cpu().numpy(), ref[k], f"{mode} {w}x{h}: pair {p} level {k} vs the oracle")


def test_config4_eight_logical_ranks_local_corner_stream(eng):
    """BASELINE config 4 (3840x2160 row-sharded over 8 ranks) the way bench.py --gpus 8 runs it by default: every rank
    streams its row block, eight frames per launch, frames read in place, shift vectors from its own top-left patch.  The
    eight ranks share the one device here; their row blocks put together must be the unsharded plain sequence."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win, R, B, nf = 3840, 2160, 5, 9, 8, 8, 19
    frames = [synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1] for i in range(nf)]
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    want = _plain_sequence(eng, d_frames, w, h, L, win, "lk_float")
    del d_frames
    _, views = _ring(frames, w, h, 28, 0, 0)
    ranks = [eng.Session(w, h, L, win, "lk_float", shard=ShardPlan(w, h, L, win, r, R), local_corner=True, stream_batch=B,
                         borrow_frames=True) for r in range(R)]
    got = _run_stream(eng, ranks, views, frames, B, L, w)
    assert sorted(got) == list(range(1, nf))
    for p in range(1, nf):
        for k in range(L):
            full = torch.cat([got[p][r][k] for r in range(R)], dim=0)
            assert _same_bits(full, want[p][k]), f"8 ranks: pair {p} level {k}"
    for s in ranks:
        assert s.corner_status() == 0
        s.close()


@pytest.mark.parametrize("mode", ["lk_float", "compat_cpu"])
def test_config4_eight_logical_ranks_halo_exchange(eng, mode):
    """BASELINE config 4 in north_star's literal formulation: every rank is handed ONLY its own rows of the 4K frame (the
    rest of its buffer is poison), downsamples its own rows and receives the halo rows of every pyramid level from its
    neighbours -- device copies here, where ShardedFlow.exchange_halos posts the RCCL send/recv pairs
    (tests/test_parallel.py runs that code under gloo) -- then rank 0's corner kernel, the copy that stands in for the
    broadcast, and the LK launch.  Bit-exact against the unsharded session."""
    import torch
    from cuda_optical_flow_2_amd.parallel import HipBackend, ShardPlan

    w, h, L, win, R = 3840, 2160, 5, 9, 8
    frames = [synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1] for i in range(3)]
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
            got = torch.cat([b.flow(k) for b in ranks], dim=0)
            assert _same_bits(got, whole.flow(k)[0]), f"{mode} frame {i} level {k}"
        whole.swap()
        for b in ranks:
            b.swap()
    whole.close()
    for b in ranks:
        b.session.close()


@pytest.mark.parametrize("cfg", [(1920, 1080, 4, 7, 5), (3840, 2160, 5, 9, 5)])
def test_literal_baseline_configs_with_iterations_match_oracle(eng, oracle, cfg):
    """BASELINE configs 2 and 3 as written ("5 iters"): the lk_iter extension at full size against
    its CPU restatement (orc_lk_iter_level), whole frames, every level, bit for bit."""
    w, h, L, win, iters = cfg
    p, n = synth.smooth_pair(w, h)
    got = eng.flow_pair(p, n, L, win, "lk_float", iters=iters)
    want = oracle.flow_pair_iter(p, n, L, win, iters)
    for k in range(L):
        assert_same(got[k], want[k], f"{w}x{h} iters={iters} level {k}")


def test_config5_8k_eight_logical_ranks_and_streamed_iterations(eng, oracle):
    """BASELINE config 5 (7680x4320, 6 levels, 15x15, 10 iterations, 8 GPUs) on the one device there is:
    (1) the pair row-sharded over 8 logical ranks, every rank streaming its block with local corner flows (four frames per
        launch: 8 x 6 levels would exceed the 40 items of a launch), put together == the unsharded plain sequence;
    (2) its 10 refinement iterations through the stream pipeline == the pair-at-a-time path, every pair and level, bit for
        bit -- unsharded, and then ALL of config 5 at once: 8 logical ranks, each streaming its block with 10 iterations
        (the ranks' blocks put together == the same unsharded result) (that path is tied to the oracle's orc_lk_iter_level at 1080p / 4K above and at small sizes with up to 5 levels
        in test_gpu_parity.py; the oracle needs minutes for an 8K pair with 10 iterations);
    (3) a 1/16-area crop of the same frames (1920x1088, the whole pipeline: 6 levels, 15x15, 10 iterations) against the
        oracle."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardPlan

    w, h, L, win = 7680, 4320, 6, 15
    nf = 7
    frames = [synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1] for i in range(nf)]
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    want = _plain_sequence(eng, d_frames, w, h, L, win, "lk_float")
    R, B = 8, 4
    _, views = _ring(frames, w, h, 16, 0, 0)
    ranks = [eng.Session(w, h, L, win, "lk_float", shard=ShardPlan(w, h, L, win, r, R), local_corner=True, stream_batch=B,
                         borrow_frames=True) for r in range(R)]
    got = _run_stream(eng, ranks, views, frames, B, L, w)
    for p in range(1, nf):
        for k in range(L):
            assert _same_bits(torch.cat([got[p][r][k] for r in range(R)], dim=0), want[p][k]), f"8K, 8 ranks: pair {p} level {k}"
    for s in ranks:
        assert s.corner_status() == 0
        s.close()
    del got, want, ranks

    iters = 10
    want = _plain_sequence(eng, d_frames[:4], w, h, L, win, "lk_float", iters=iters)
    s = eng.Session(w, h, L, win, "lk_float", iters=iters, stream_batch=2)
    got = _run_stream(eng, [s], [v for v in d_frames[:4]], frames[:4], 2, L, w)
    s.close()
    for p in range(1, 4):
        for k in range(L):
            assert _same_bits(got[p][0][k], want[p][k]), f"8K iters={iters}: streamed pair {p} level {k} != pair-at-a-time"
    del got
    ranks = [eng.Session(w, h, L, win, "lk_float", shard=ShardPlan(w, h, L, win, r, R, iters=iters, warp_margin=16), local_corner=True,
                         stream_batch=2, iters=iters) for r in range(R)]
    got = _run_stream(eng, ranks, [v for v in d_frames[:4]], frames[:4], 2, L, w)
    status = [s.corner_status() for s in ranks]
    for s in ranks:
        s.close()
    assert status == [0] * R, [hex(x) for x in status]
    for p in range(1, 4):
        for k in range(L):
            assert _same_bits(torch.cat([got[p][r][k] for r in range(R)], dim=0), want[p][k]), f"8K, 8 ranks, iters={iters}: pair {p} level {k}"
    del got, want, d_frames, ranks

    cw, ch = 1920, 1088   # (even at every level that is downsampled; 1080 >> 3 = 135 is not)
    a, b = frames[0][:ch, :cw].copy(), frames[1][:ch, :cw].copy()
    hip = eng.flow_pair(a, b, L, win, "lk_float", iters=iters)
    ref = oracle.flow_pair_iter(a, b, L, win, iters)
    for k in range(L):
        assert_same(hip[k], ref[k], f"crop 1920x1088, 6 levels, 15x15, iters={iters}: level {k}")
