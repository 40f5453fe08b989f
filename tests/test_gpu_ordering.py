"""GPU: ordering and status contracts of the session that the parity tests (static, pre-synchronised frames) cannot see."""
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


def _busy(torch, a):
    """a few milliseconds of work on the current stream"""
    for _ in range(6):
        a = a @ a
        a = a / a.abs().max()
    return a


def test_submit_device_waits_for_the_frames_producer(eng):
    """ofx_session_submit_device reads the frame on the session's own staging stream.  The frame here is written by the
    caller's stream BEHIND several milliseconds of other work and poisoned again right after the submit, with no host
    synchronisation anywhere: the staging stream must be ordered after the producer (and the caller's next write after the
    staging copy), or the flow is computed from poison."""
    import torch

    w, h, L, win, nf = 1280, 720, 4, 9, 6
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.7 * i, -0.9 * i, seed=5)[1]).cuda() for i in range(nf)]
    plain = eng.Session(w, h, L, win, "lk_float")
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    want = []
    for i in range(1, nf):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        torch.cuda.synchronize()
        want.append([plain.flow_host(k) for k in range(L)])
        plain.swap()
    plain.close()

    piped = eng.Session(w, h, L, win, "lk_float")
    piped.set_frame_device(frames[0]); piped.build_pyramid(); piped.swap()
    buf = torch.full((h, w), 0xEE, dtype=torch.uint8, device="cuda")
    a = torch.rand((2048, 2048), device="cuda")
    torch.cuda.synchronize()
    snaps = []
    for i in range(1, nf):
        a = _busy(torch, a)
        buf.copy_(frames[i])            # the producer: runs only once the matmuls above have drained
        piped.submit_device(buf)
        buf.fill_(0xEE)                 # reuse of the surface, in stream order behind the pair's LK launch
        snaps.append([piped.flow(k)[0].clone() for k in range(L)])
    torch.cuda.synchronize()
    for i, (got, ref) in enumerate(zip(snaps, want)):
        for k in range(L):
            assert_same(got[k].cpu().numpy(), ref[k], f"pair {i + 1} level {k}")
    piped.close()


def test_sharded_driver_orders_its_staging_stream_after_the_producer(eng):
    """The same through parallel.ShardedFlow.step (HipBackend.pipelined_step enters the session's staging stream)."""
    import torch
    from cuda_optical_flow_2_amd.parallel import ShardedFlow

    w, h, L, win = 1280, 720, 3, 7
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.1 * i, 0.6 * i, seed=9)[1]).cuda() for i in range(5)]
    plain = eng.Session(w, h, L, win, "lk_float")
    plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
    drv = ShardedFlow(w, h, L, win, "lk_float", 0, 1)
    drv.push_frame(frames[0])
    buf = torch.full((h, w), 0xEE, dtype=torch.uint8, device="cuda")
    a = torch.rand((2048, 2048), device="cuda")
    torch.cuda.synchronize()
    for i in range(1, 5):
        plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
        a = _busy(torch, a)
        buf.copy_(frames[i])
        drv.step(buf)
        buf.fill_(0xEE)
        torch.cuda.synchronize()
        for k in range(L):
            assert_same(drv.gather_flow(k).cpu().numpy(), plain.flow_host(k), f"pair {i} level {k}")
        plain.swap()
    assert drv.corner_status() == 0
    plain.close()
    drv.session.close()


def test_broadcast_mode_reports_a_shift_beyond_the_halo(eng):
    """Sharded sessions that RECEIVE their shift vectors (rank 0's corner kernel + broadcast) check them on the device in
    ofx_session_run_levels: a vertical shift that sends the shard's reads to image rows its buffers do not hold raises bit
    8 + level of the status word -- the same bit local_corner sessions raise -- without a host synchronisation."""
    import torch
    from cuda_optical_flow_2_amd.parallel import HipBackend, ShardPlan

    w, h, L, win, R = 640, 480, 3, 9, 3
    frames = [torch.from_numpy(synth.smooth_pair(w, h, 1.5 * i, 0.75 * i, seed=3)[1]).cuda() for i in range(2)]
    ranks = [HipBackend(ShardPlan(w, h, L, win, r, R), "lk_float", 0) for r in range(R)]
    for b in ranks:
        b.load_frame(frames[0]); b.build_pyramid(); b.swap()
        b.load_frame(frames[1]); b.build_pyramid()
    ranks[0].corner_flows()
    torch.cuda.synchronize()
    uv = ranks[0].uv_all.clone()
    # (1) the real vectors of this smooth pair are tiny: nobody flags anything
    for b in ranks:
        b.uv_all.copy_(uv)
        b.run_levels()
    assert [b.session.corner_status() for b in ranks] == [0, 0, 0]
    # (2) a level-1 shift of -30 rows (the plan's margin is 8): ranks 1 and 2 would read rows above their buffers
    bad = uv.clone()
    bad[2 * 1 + 1] = -30.0
    for b in ranks:
        b.uv_all.copy_(bad)
        b.run_levels()
    st = [b.session.corner_status() for b in ranks]
    assert st[0] == 0 and all((s >> 8) == 0b10 for s in st[1:]), [hex(s) for s in st]
    # (3) a NaN shift moves nothing (OptFlowCPU.cpp:270: every target is out of range) and is not an error
    nan = uv.clone()
    nan[2 * 0 + 1] = float("nan")
    for b in ranks:
        b.uv_all.copy_(nan)
        b.run_levels()
    assert [b.session.corner_status() for b in ranks] == [0, 0, 0]
    for b in ranks:
        b.session.close()


def test_timing_kinds_cover_every_launch_of_a_pair(eng, monkeypatch):
    """ofx_session_timing tags every launch of the plain path; iters > 1 adds the shift and the accumulating launches -- and no warp
    launch: every launch but the last writes the warped image of the iteration after it (csrc/lk_body_warp.h); OFX_ITER_FUSED=0
    keeps one warp launch per refinement iteration."""
    import torch

    w, h, L, win = 640, 480, 3, 9
    p, n = synth.smooth_pair(w, h, 1.0, 0.5)
    # (lk_acc_warp: an accumulating launch that also writes the next iteration's warped image -- every one but the last)
    for iters, fused, want in ((1, "1", {"pyramid": 1, "corner": 1, "lk": 1, "lk_acc": 0, "lk_acc_warp": 0, "warp": 0, "shift": 0}),
                               (3, "1", {"pyramid": 1, "corner": 1, "lk": 1, "lk_acc": 1, "lk_acc_warp": 1, "warp": 0, "shift": 1}),
                               (3, "0", {"pyramid": 1, "corner": 1, "lk": 1, "lk_acc": 2, "lk_acc_warp": 0, "warp": 2, "shift": 1})):
        monkeypatch.setenv("OFX_ITER_FUSED", fused)
        s = eng.Session(w, h, L, win, "lk_float", iters=iters)
        s.push_frame_host(p)
        s.timing(16)
        s.set_frame_host(n); s.build_pyramid(); s.run_flow()
        torch.cuda.synchronize()
        for kind, count in want.items():
            avg, mn, cnt = s.timing_read_kind(kind)
            assert cnt == count, (iters, kind, cnt)
            assert count == 0 or (0 < mn <= avg < 1e5)
        avg, mn, cnt = s.timing_read()
        assert cnt == want["lk"] + want["lk_acc"] + want["lk_acc_warp"]
        s.close()
