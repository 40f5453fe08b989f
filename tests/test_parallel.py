"""CPU: the row-sharding plan and the N > 1 driver (cuda_optical_flow_2_amd/parallel.py) under gloo, world_size 2 and 3.

The per-rank compute here is an ORACLE-backed stand-in for the HIP session (same interface, numpy + oracle calls), so
the test exercises exactly what multi-GPU adds: the partition, the halo/recompute ranges, the broadcast of the shift
vectors and the reassembly -- and checks the sharded result against the unsharded oracle bit for bit.
"""
import os
import socket

import numpy as np
import pytest

from cuda_optical_flow_2_amd import synth
from cuda_optical_flow_2_amd.parallel import ShardPlan, ShardedFlow


# ---- plan invariants -------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("cfg", [(3840, 2160, 5, 9, 8), (1920, 1080, 4, 7, 4), (7680, 4320, 6, 15, 8), (640, 480, 3, 5, 2),
                                 (256, 192, 3, 9, 3), (64, 48, 1, 5, 2)])
def test_plan_invariants(cfg):
    w, h, L, win, world = cfg
    plans = [ShardPlan(w, h, L, win, r, world) for r in range(world)]
    for k in range(L):
        hk = h >> k
        cuts = [p.own[k] for p in plans]
        assert cuts[0][0] == 0 and cuts[-1][1] == hk
        assert all(a[1] == b[0] for a, b in zip(cuts[:-1], cuts[1:])), "own blocks must tile the level"
        for p in plans:
            o, n, c, b = p.own[k], p.need[k], p.comp[k], p.buf[k]
            assert b[0] <= n[0] <= o[0] <= o[1] <= n[1] <= b[1] and 0 <= b[0] < b[1] <= hk
            assert b[0] <= c[0] <= o[0] and o[1] <= c[1] <= b[1]
            halo = win // 2 + 1 + p.margin
            assert n[0] == max(0, o[0] - halo) and n[1] == min(hk, o[1] + halo)
            if k + 1 < L:  # everything level k+1 computes is a 3-row stencil of rows this rank holds at level k
                c1 = p.comp[k + 1]
                assert b[0] <= max(0, 2 * c1[0] - 1) and min(hk, 2 * c1[1] + 1) <= b[1]
    if cfg[0] == 3840:
        # 4K over 8 ranks: 135 coarse rows -> 17 x 7 + 16, level-0 blocks of 272 / 256 rows
        assert [p.own[4][1] - p.own[4][0] for p in plans] == [17] * 7 + [16]
        assert [p.own[0][1] - p.own[0][0] for p in plans] == [272] * 7 + [256]


def test_plan_rejects_impossible_splits():
    with pytest.raises(ValueError):
        ShardPlan(64, 48, 4, 5, 0, 8)       # 6 coarse rows over 8 ranks
    with pytest.raises(ValueError):
        ShardPlan(100, 50, 3, 5, 0, 2)      # odd level gets downsampled


# ---- oracle-backed rank ---------------------------------------------------------------------------------------------

class OracleBackend:
    """numpy/oracle stand-in for HipBackend: one rank's buffers hold only the rows plan.buf says they hold."""

    def __init__(self, plan, mode, patch_size=0):
        import torch
        from oracle import Oracle

        self.plan, self.mode, self.orc = plan, mode, Oracle()
        # "local" corner mode: the top-left patch of every frame as a pyramid of its own (side as in ofx_session_create)
        step = 1 << (plan.levels - 1)
        side = patch_size if patch_size > 0 else max(256, step * (plan.window // 2 + 2 + 8))
        side = -(-side // step) * step
        self.patch_wh = (min(side, plan.width), min(side, plan.height))
        self.patch_prev, self.patch_next = None, None
        L = plan.levels
        self.w = [plan.width >> k for k in range(L)]
        self.h = [plan.height >> k for k in range(L)]
        self.prev, self.next = [None] * L, [None] * L
        self.flows = [np.zeros((plan.own[k][1] - plan.own[k][0], self.w[k], 2), np.float32) for k in range(L)]
        self.uv_all = torch.zeros(2 * L, dtype=torch.float32)

    def load_frame(self, frame):
        b0, b1 = self.plan.buf[0]
        self.next[0] = np.asarray(frame)[b0:b1].copy()
        pw, ph = self.patch_wh
        pyr = [np.asarray(frame)[:ph, :pw].copy()]
        for k in range(1, self.plan.levels):   # the patch is downsampled like an image of its own
            pyr.append(self.orc.downscale_gaussian(synth.to_3ch(pyr[-1]))[:, :, 0].copy())
        self.patch_next = pyr

    def corner_flows_local(self):
        """The corner chain from this rank's own patch pyramids: no rank owns anything the others need."""
        L, r = self.plan.levels, self.plan.window // 2
        f0 = {}
        for k in range(L - 1, -1, -1):
            u = v = np.float32(0)
            for j in range(L - 1, k, -1):
                m = np.float32(1 << (j - k))
                u = np.float32(u + m * f0[j][0])
                v = np.float32(v + m * f0[j][1])
            w, h = self.w[k], self.h[k]
            rows, cols = min(h, r + 2), min(w, r + 2)
            prev, nxt_src = self.patch_prev[k], self.patch_next[k]
            assert prev.shape[0] >= rows and prev.shape[1] >= cols
            if k != L - 1:
                self.uv_all[2 * k], self.uv_all[2 * k + 1] = float(u), float(v)
                nxt = np.zeros((rows, cols), np.uint8)
                for y in range(rows):
                    for x in range(cols):
                        tx, ty = np.float32(x) + u, np.float32(y) + v
                        if tx > -1 and tx < w and ty > -1 and ty < h:   # OptFlowCPU.cpp:270-276
                            nx, ny = int(np.trunc(tx)), int(np.trunc(ty))
                            assert ny < nxt_src.shape[0] and nx < nxt_src.shape[1], "shift target outside the patch"
                            nxt[y, x] = nxt_src[ny, nx]
                        else:
                            nxt[y, x] = nxt_src[y, x] if 3 * (y * w + x) < w * h else 0
            else:
                nxt = nxt_src[:rows, :cols]
            # pixel 0's window reaches column/row r and its derivatives r + 1: the (r+2)^2 crop is all it sees
            f0[k] = self._level(prev[:rows, :cols], nxt)[0, 0]

    def build_pyramid(self):
        p = self.plan
        for k in range(1, p.levels):
            c0, c1 = p.comp[k]
            assert (c0, c1) == p.buf[k]
            s0, s1 = max(0, 2 * c0 - 2), 2 * c1              # even-aligned source crop; its first row may be unused
            src0 = p.buf[k - 1][0]
            crop = np.zeros((s1 - s0, self.w[k - 1]), np.uint8)
            lo = max(s0, src0)
            assert s1 <= p.buf[k - 1][1] and lo <= max(0, 2 * c0 - 1), "plan does not hold the rows the stencil needs"
            crop[lo - s0:] = self.next[k - 1][lo - src0: s1 - src0]
            out = self.orc.downscale_gaussian(synth.to_3ch(crop))[:, :, 0]
            self.next[k] = out[c0 - s0 // 2:].copy()
            assert self.next[k].shape[0] == c1 - c0

    def downsample_level(self, k):
        """Exchange mode: this rank's OWN rows of level k from level k-1 (own rows + the halo row above them)."""
        p = self.plan
        c0, c1 = p.comp[k]
        assert (c0, c1) == p.own[k]
        s0, s1 = max(0, 2 * c0 - 2), 2 * c1
        src0 = p.buf[k - 1][0]
        crop = np.zeros((s1 - s0, self.w[k - 1]), np.uint8)
        lo = max(s0, src0)
        assert s1 <= p.buf[k - 1][1] and lo <= max(0, 2 * c0 - 1)
        crop[lo - s0:] = self.next[k - 1][lo - src0: s1 - src0]
        out = self.orc.downscale_gaussian(synth.to_3ch(crop))[:, :, 0]
        b0, b1 = p.buf[k]
        if self.next[k] is None:
            self.next[k] = np.full((b1 - b0, self.w[k]), 0xEE, np.uint8)   # halo rows: poison until the exchange fills them
        self.next[k][c0 - b0: c1 - b0] = out[c0 - s0 // 2:]

    def next_plane(self, k):
        import torch

        return torch.from_numpy(self.next[k]), self.plan.buf[k][0]

    def _shift_rows(self, k, u, v, y0, y1):
        """cpu::shift_back_pyramid on channel 0 for global rows [y0,y1) of level k, from this rank's buffer only."""
        w, h, b0 = self.w[k], self.h[k], self.plan.buf[k][0]
        src = self.next[k]
        u, v = np.float32(u), np.float32(v)
        out = np.zeros((y1 - y0, w), np.uint8)
        xs = np.arange(w)
        tx = xs.astype(np.float32) + u
        xin = (tx > -1) & (tx < w)
        nx = np.where(xin, np.trunc(np.where(xin, tx, 0)), 0).astype(np.int64)
        for y in range(y0, y1):
            ty = np.float32(y) + v
            keep = 3 * (y * w + xs) < w * h
            fallback = np.where(keep, src[y - b0], 0)
            if ty > -1 and ty < h:
                ny = int(np.trunc(ty))
                assert b0 <= ny < b0 + src.shape[0], "shift target outside the halo margin"
                out[y - y0] = np.where(xin, src[ny - b0, nx], fallback)
            else:
                out[y - y0] = fallback
        return out

    def _level(self, prev_crop, next_crop):
        fl = [np.zeros(prev_crop.shape + (2,), np.float32)]
        p3, n3 = synth.to_3ch(prev_crop), synth.to_3ch(next_crop)
        if self.mode == "compat_cpu":
            self.orc.calc_optical_flow_cpu(p3, n3, fl, 0, 1, self.plan.window)
        else:
            self.orc.calc_opt_flow_gpu(p3, n3, fl, 0, 1, self.plan.window, exact_sums=True)
        return fl[0]

    def corner_flows(self):
        p, L, r = self.plan, self.plan.levels, self.plan.window // 2
        assert p.rank == 0
        f0 = {}
        for k in range(L - 1, -1, -1):
            u = v = np.float32(0)
            for j in range(L - 1, k, -1):
                m = np.float32(1 << (j - k))
                u = np.float32(u + m * f0[j][0])
                v = np.float32(v + m * f0[j][1])
            rows, cols = min(self.h[k], r + 2), min(self.w[k], r + 2)
            if k != L - 1:
                self.uv_all[2 * k], self.uv_all[2 * k + 1] = float(u), float(v)
                nxt = self._shift_rows(k, u, v, 0, rows)
            else:
                nxt = self.next[k][:rows]
            f0[k] = self._level(self.prev[k][:rows, :cols], nxt[:, :cols])[0, 0]

    def run_levels(self):
        p, L, halo = self.plan, self.plan.levels, self.plan.window // 2 + 1
        for k in range(L):
            o0, o1 = p.own[k]
            y0, y1 = max(0, o0 - halo), min(self.h[k], o1 + halo)
            b0 = p.buf[k][0]
            if k != L - 1:
                nxt = self._shift_rows(k, float(self.uv_all[2 * k]), float(self.uv_all[2 * k + 1]), y0, y1)
            else:
                nxt = self.next[k][y0 - b0: y1 - b0]
            self.flows[k] = self._level(self.prev[k][y0 - b0: y1 - b0], nxt)[o0 - y0: o1 - y0].copy()

    def swap(self):
        self.prev, self.next = self.next, [None] * self.plan.levels
        self.patch_prev, self.patch_next = self.patch_next, None

    def flow(self, level):
        import torch

        return torch.from_numpy(self.flows[level])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, cfg, mode, corner="broadcast", halo_mode="recompute"):
    import torch.distributed as dist
    from conftest import assert_same
    from oracle import Oracle

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h, L, win, margin = cfg
        frames = [synth.smooth_pair(w, h, 1.5 * i, 0.75 * i, seed=5)[1] for i in range(3)]
        plan = ShardPlan(w, h, L, win, rank, world, margin, halo_mode)
        sf = ShardedFlow(w, h, L, win, mode, rank, world, margin=margin, backend=OracleBackend(plan, mode, patch_size=48),
                         corner=corner, halo_mode=halo_mode)

        def seen_by_rank(frame):
            """Exchange mode: a rank only has its own rows of a frame; everything else is poison it must never use."""
            if halo_mode != "exchange":
                return frame
            f = np.full_like(frame, 0xEE)
            o0, o1 = plan.own[0]
            f[o0:o1] = frame[o0:o1]
            return f

        import torch

        def own_rows(frame):   # stream_exchange: all a rank is ever given of a frame
            o0, o1 = plan.own[0]
            return torch.from_numpy(frame[o0:o1].copy())

        if halo_mode == "stream_exchange":
            sf._frame_buffers(1, own_rows(frames[0]))
            sf._rings[1][0].fill_(0xEE)   # whatever the exchange does not bring stays poison
            sf.push_own_rows(own_rows(frames[0]))
        else:
            sf.push_frame(seen_by_rank(frames[0]))
        orc = Oracle()
        for i in (1, 2):
            if halo_mode == "stream_exchange":
                sf.step_own_rows(own_rows(frames[i]))
            else:
                sf.step(seen_by_rank(frames[i]), check_margin=corner == "broadcast")
            want, _, _ = orc.flow_pair(synth.to_3ch(frames[i - 1]), synth.to_3ch(frames[i]), L, win, mode, exact_sums=True)
            for k in range(L):
                got = sf.gather_flow(k).numpy()
                assert_same(got, want[k], f"rank {rank}/{world} frame {i} level {k}")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["lk_float", "compat_cpu"])
def test_sharded_flow_gloo(world, mode):
    import torch.multiprocessing as mp

    cfg = (96, 144, 3, 5, 8)  # 36 coarse rows; window 5; halo margin 8
    mp.spawn(_rank_main, args=(world, _free_port(), cfg, mode), nprocs=world, join=True)


@pytest.mark.parametrize("mode", ["lk_float", "compat_cpu"])
def test_sharded_flow_gloo_local_corner(mode):
    """corner="local": every rank forms the shift vectors from its own 48x48 top-left patch pyramid; no broadcast.  The
    collective left in the test is gather_flow's all_gather, which only reassembles the result for the comparison."""
    import torch.multiprocessing as mp

    cfg = (96, 144, 3, 5, 8)
    mp.spawn(_rank_main, args=(2, _free_port(), cfg, mode, "local"), nprocs=2, join=True)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["lk_float", "compat_cpu"])
def test_sharded_flow_gloo_halo_exchange(world, mode):
    """halo_mode="exchange" (north_star's formulation): a rank is given ONLY its own rows of each frame (the rest is
    poisoned), downsamples only its own rows, and fetches the halo rows of every pyramid level from its neighbours with
    one batched send/recv per neighbour and level; rank 0's shift vectors are broadcast.  Bit-exact against the
    unsharded oracle."""
    import torch.multiprocessing as mp

    cfg = (96, 288, 3, 5, 4)  # 72 coarse rows: every rank owns >= the halo (3 + 4) rows at every level, also with 3 ranks
    mp.spawn(_rank_main, args=(world, _free_port(), cfg, mode, "broadcast", "exchange"), nprocs=world, join=True)


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("mode", ["lk_float", "compat_cpu"])
def test_sharded_flow_gloo_stream_exchange(world, mode):
    """halo_mode="stream_exchange": a rank is given ONLY its own rows of each frame; one batched group of sends / receives per
    frame brings the level-0 halo rows of its plan and the rows / columns of the frame's top-left patch (from which it forms the
    shift vectors itself); the halos of the coarser levels are recomputed.  Everything else in its frame buffer is poison.
    Bit-exact against the unsharded oracle -- the form in which sharded arrivals run the one-launch stream pipeline."""
    import torch.multiprocessing as mp

    cfg = (96, 144, 3, 5, 8)
    mp.spawn(_rank_main, args=(world, _free_port(), cfg, mode, "local", "stream_exchange"), nprocs=world, join=True)


def test_stream_exchange_lists_cover_the_plan():
    """what a rank receives = exactly the rows of its level-0 buffer it does not own + the patch rows outside that buffer, each
    from the rank that owns them; what it sends mirrors what the others receive"""
    for (w, h, L, win, world) in ((3840, 2160, 5, 9, 8), (1920, 1080, 4, 7, 4), (96, 144, 3, 5, 3)):
        flows = [ShardedFlow(w, h, L, win, "lk_float", r, world, backend=object(), corner="local", halo_mode="stream_exchange") for r in range(world)]
        lists = [f._exchange_lists() for f in flows]
        for me in range(world):
            p = flows[me].plan
            pw, ph = p.patch_wh(0)
            got = np.zeros((h, w), bool)
            got[p.own[0][0]:p.own[0][1]] = True
            for peer, a, b, cols in lists[me][0]:
                src = flows[peer].plan.own[0]
                assert src[0] <= a < b <= src[1], "a piece must come from the rank that owns it"
                assert not got[a:b, :cols].any(), "nothing is received twice"
                got[a:b, :cols] = True
                assert (me, a, b, cols) in [(q, x, y, c) for q, x, y, c in lists[peer][1]], "every receive has its send"
            b0, b1 = p.buf[0]
            assert got[b0:b1].all() and got[:ph, :pw].all(), "the level-0 buffer rows and the patch are complete"
            assert sum(len(l[0]) for l in lists) == sum(len(l[1]) for l in lists)


def test_exchange_plan_invariants():
    for world in (2, 4, 8):
        plans = [ShardPlan(3840, 2160, 5, 9, r, world, 8, "exchange") for r in range(world)]
        for p in plans:
            assert p.comp == p.own and p.buf == p.need
    with pytest.raises(ValueError):
        ShardPlan(96, 144, 3, 5, 0, 4, 8, "exchange")   # 9 coarse rows per rank < halo 11: a neighbour's halo would span two ranks


def _rank_assemble(rank, world, port):
    """a tick of THREE frames through ShardedFlow.assemble_frames in its stacked form (one message per peer and direction): every
    buffer must hold the frame's rows of the plan and its top-left patch, and poison everywhere else"""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        w, h, L, win = 96, 144, 3, 5
        plan = ShardPlan(w, h, L, win, rank, world, 8, "stream_exchange")
        sf = ShardedFlow(w, h, L, win, "lk_float", rank, world, margin=8, backend=OracleBackend(plan, "lk_float", patch_size=48),
                         corner="local", halo_mode="stream_exchange", patch_size=48)
        frames = [synth.random_pair(w, h, 40 + i)[0] for i in range(3)]
        o0, o1 = plan.own[0]
        stack = torch.from_numpy(np.stack([f[o0:o1] for f in frames]))
        bufs = torch.full((3, h, w), 0xEE, dtype=torch.uint8)
        sf.assemble_frames(stack, bufs)
        pw, ph = plan.patch_wh(48)
        b0, b1 = plan.buf[0]
        for i, f in enumerate(frames):
            want = np.full((h, w), 0xEE, np.uint8)
            want[b0:b1] = f[b0:b1]
            want[:ph, :pw] = f[:ph, :pw]
            assert np.array_equal(bufs[i].numpy(), want), f"rank {rank}/{world} frame {i}"
        # the list form (the pair-at-a-time callers) brings the same bytes
        bufs2 = [torch.full((h, w), 0xEE, dtype=torch.uint8) for _ in frames]
        sf.assemble_frames([torch.from_numpy(f[o0:o1].copy()) for f in frames], bufs2)
        for i in range(3):
            assert torch.equal(bufs2[i], bufs[i])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_assemble_frames_of_a_tick_in_one_exchange_gloo(world):
    import torch.multiprocessing as mp

    mp.spawn(_rank_assemble, args=(world, _free_port()), nprocs=world, join=True)
