"""Row-sharded dense LK over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests).

Partition.  The COARSEST pyramid level's rows are split as evenly as possible over the ranks and the cut is scaled up
by 2 per level, so every level is cut at the same image position and a rank owns one contiguous row block per level.

What crosses ranks.  Every stage is a finite-support stencil, so a rank only ever needs rows within a fixed distance
of its block:
  * LK at level k reads (radius + 1) rows beyond the block, the shift up to `margin` rows further;
  * level k+1's rows are a 3-row stencil of level k (rows 2y-1 .. 2y+1).
In the default "recompute" mode those halo rows are not fetched from the neighbours at every level: the rank takes a
wider halo of the NEW FRAME's level 0 once (the frame arrives in every rank's HBM anyway) and rebuilds the halo rows of
the coarser levels itself -- a few per cent more downsampling work instead of a latency-bound exchange per level.  The
one true dependency between ranks is the reference's shift vector, which is formed from PIXEL 0 of every coarser flow
level (OptFlowCPU.cpp:255-266): rank 0 owns that corner, runs the corner kernel, and broadcasts the 2*levels floats --
one small collective per frame pair ("broadcast" corner mode).

"local" corner mode removes that collective too.  Pixel 0's flow only depends on the top-left (radius + 2 + shift)
pixels of every level, and the pyramid's stencil (2x-1 .. 2x+1) never reaches past column/row 2*w_k - 1, so the pyramid
of a frame's top-left PATCH equals the top-left part of the frame's pyramid at every level.  Every rank is handed the
whole frame anyway, so each rank builds that small patch pyramid next to its row block and runs the corner chain
itself: ranks share nothing, and each of them can run the one-launch-per-frame stream pipeline
(ofx_session_stream_*).  The shift is exact while it stays inside the patch (the session reports when it does not).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

Range = Tuple[int, int]


def _clip(lo: int, hi: int, n: int) -> Range:
    return max(0, lo), min(n, hi)


@dataclass
class ShardPlan:
    """Row ranges of one rank at every pyramid level (all global row indices, half-open)."""

    width: int
    height: int
    levels: int
    window: int
    rank: int
    world: int
    margin: int = 8  # rows of slack for the reference's global shift (|v| <= margin keeps the shift exact)
    # "recompute": halos rebuilt from a wider level-0 halo; "exchange": fetched from the neighbours at every level;
    # "stream_exchange": the ranges of "recompute" -- only level-0 rows ever cross ranks (ShardedFlow.assemble_frames)
    halo_mode: str = "recompute"
    # refinement iterations (extension, ofx_params.iters): iteration j is computed on (radius + 1) * (iters - j) rows beyond the
    # block, so that the warp of iteration j + 1 finds the flow of every row it needs without a neighbour; the warp itself may
    # reach `warp_margin` rows further (|scale * v| + 1 <= warp_margin keeps it exact; the session reports when it did not)
    iters: int = 1
    warp_margin: int = 8
    own: List[Range] = field(default_factory=list)   # rows whose flow this rank computes
    need: List[Range] = field(default_factory=list)  # rows of the images LK + shift may touch
    comp: List[Range] = field(default_factory=list)  # rows this rank downsamples itself (levels >= 1)
    buf: List[Range] = field(default_factory=list)   # rows its plane buffers hold

    def __post_init__(self):
        L = self.levels
        hs = [self.height >> k for k in range(L)]
        for k in range(L - 1):
            if hs[k] % 2 or (self.width >> k) % 2:
                raise ValueError(f"level {k} ({self.width >> k}x{hs[k]}) must have even dimensions to be downsampled")
        hc = hs[L - 1]
        if hc < self.world:
            raise ValueError(f"cannot split {hc} coarsest-level rows over {self.world} ranks")
        base, extra = divmod(hc, self.world)
        c0 = self.rank * base + min(self.rank, extra)
        c1 = c0 + base + (1 if self.rank < extra else 0)
        halo = self.window // 2 + 1 + self.margin
        if self.iters > 1:
            halo += (self.iters - 1) * (self.window // 2 + 1) + self.warp_margin
        self.own = [(c0 << (L - 1 - k), c1 << (L - 1 - k)) for k in range(L)]
        self.need = [_clip(self.own[k][0] - halo, self.own[k][1] + halo, hs[k]) for k in range(L)]
        # recompute mode: walk down from the coarsest level; what level k+1 computes dictates what level k must hold
        self.comp = [None] * L
        self.buf = [None] * L
        self.comp[L - 1] = self.need[L - 1]
        self.buf[L - 1] = self.need[L - 1]
        for k in range(L - 2, -1, -1):
            src = _clip(2 * self.comp[k + 1][0] - 1, 2 * self.comp[k + 1][1] + 1, hs[k])  # rows 2y-1 .. 2y+1
            lo, hi = min(self.need[k][0], src[0]), max(self.need[k][1], src[1])
            self.buf[k] = (lo, hi)
            self.comp[k] = (lo, hi)
        if L == 1:
            self.comp[0] = self.buf[0] = self.need[0]
        if self.halo_mode == "exchange":
            # every level holds exactly what LK needs; a rank downsamples its own rows only and the halo rows of every
            # level come from the neighbouring ranks (north_star's "halo exchange at each pyramid level")
            self.buf = list(self.need)
            self.comp = list(self.own)
            for k in range(L):
                rows = self.own[k][1] - self.own[k][0]
                if self.world > 1 and rows < halo:
                    raise ValueError(f"level {k}: a rank owns {rows} rows, fewer than the halo of {halo}: its neighbours would need rows "
                                     "from two ranks away (use fewer ranks or the recompute mode)")
        elif self.halo_mode not in ("recompute", "stream_exchange"):
            raise ValueError(f"halo_mode {self.halo_mode!r}")

    def patch_wh(self, patch_size: int = 0) -> Tuple[int, int]:
        """(width, height) of the top-left patch the corner chain reads at level 0: ofx_session_create's rule."""
        step = 1 << (self.levels - 1)
        side = patch_size if patch_size > 0 else max(256, step * (self.window // 2 + 2 + 8))
        side = -(-side // step) * step
        return min(side, self.width), min(side, self.height)

    def redundancy(self) -> float:
        """Fraction of extra level-0 rows held beyond the owned block (the price of exchanging nothing per level)."""
        own = self.own[0][1] - self.own[0][0]
        return (self.buf[0][1] - self.buf[0][0]) / max(1, own) - 1.0


class HipBackend:
    """The device-resident session of libofx_hip.so behind the operations ShardedFlow needs."""

    def __init__(self, plan: ShardPlan, mode: str, device: int, local_corner: bool = False, patch_size: int = 0,
                 stream_batch: int = 1, borrow_frames: bool = False):
        from . import engine

        self.plan = plan
        # "stream_exchange": a rank's frame buffers hold its plan's rows and the top-left patch, nothing else -- the session must
        # not repair a corner shift from rows that never arrived (ofx_params.frames_partial: the status bit stays an error)
        self.session = engine.Session(plan.width, plan.height, plan.levels, plan.window, mode, device=device, shard=plan,
                                      local_corner=local_corner, patch_size=patch_size, stream_batch=stream_batch,
                                      borrow_frames=borrow_frames, iters=plan.iters,
                                      frames_partial=plan.halo_mode == "stream_exchange")
        self._views = {}
        # the collective runs on a torch-allocated staging tensor (RCCL then only ever sees caching-allocator memory)
        self.uv_stage = self.uv_all.new_zeros(self.uv_all.shape)

    @property
    def uv_all(self):
        """View over the 2*levels shift-vector floats of the pair in progress (the session alternates two slots)."""
        from . import engine

        ptr = self.session.uv(0).data_ptr()
        v = self._views.get(ptr)
        if v is None:
            v = self._views[ptr] = engine.DeviceView(ptr, (2 * self.plan.levels,), "<f4").tensor()
        return v

    def load_frame(self, frame):
        self.session.set_frame_device(frame)

    def pipelined_step(self, frame, rank: int, world: int):
        """One pair with the staging half (frame load, pyramid, corner flows, broadcast, shifts) on the session's
        auxiliary stream, so that it runs underneath the previous pair's LK launch; the solve half goes on the current
        stream.  Same results as load_frame/build_pyramid/corner_flows/broadcast/run_levels/swap."""
        import torch
        import torch.distributed as dist

        if not hasattr(self, "_aux"):
            self._aux_ptr = self.session.aux_stream_ptr()
            self._aux = torch.cuda.ExternalStream(self._aux_ptr)
        s = self.session
        # the frame may still be in production on the caller's stream (async upload, decoder, grayscale kernel): the
        # staging stream reads it, so it is ordered behind everything enqueued there so far
        self._aux.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._aux):
            s.stage_frame(frame, self._aux_ptr)
            if rank == 0:
                s.corner_flows(self._aux)
            if world > 1 or (dist.is_available() and dist.is_initialized()):
                if rank == 0:
                    self.uv_stage.copy_(self.uv_all)
                dist.broadcast(self.uv_stage, src=0)   # issued against the current (= aux) stream
                self.uv_all.copy_(self.uv_stage)
            s.stage_shift(self._aux_ptr)
        s.solve_staged()

    def build_pyramid(self):
        self.session.build_pyramid()

    def downsample_level(self, level: int):
        self.session.downsample_level(level)

    def next_plane(self, level: int):
        """(tensor [buffer rows, >= width], global index of its first row) of the frame being loaded."""
        t, g = self.session.plane(1, level)
        return t, g.row0

    def corner_flows(self):
        self.session.corner_flows()

    def run_levels(self):
        self.session.run_levels()

    def swap(self):
        self.session.swap()

    def flow(self, level: int):
        return self.session.flow(level)[0]


class ShardedFlow:
    """One frame pair per step(), row-sharded over the ranks of the default process group."""

    def __init__(self, width, height, levels, window, mode, rank, world, device=0, margin=8, backend=None, pipelined=True,
                 corner="broadcast", patch_size=0, stream_batch=1, halo_mode="recompute", borrow_frames=False, iters=1):
        assert corner in ("broadcast", "local")
        assert halo_mode in ("recompute", "stream_exchange") or corner == "broadcast", "the exchange mode is pair-at-a-time (rank 0's corner + broadcast)"
        assert halo_mode != "stream_exchange" or corner == "local", "stream_exchange: every rank forms the shift vectors from the patch rows it receives"
        self.patch_size = patch_size
        assert iters <= 1 or corner == "local", "refinement iterations on a sharded pair run through the stream pipeline (corner='local')"
        self.plan = ShardPlan(width, height, levels, window, rank, world, margin, halo_mode, iters=iters)
        self.rank, self.world, self.pipelined, self.corner = rank, world, pipelined, corner
        self.backend = backend if backend is not None else HipBackend(self.plan, mode, device, corner == "local", patch_size,
                                                                      stream_batch, borrow_frames)
        self.session = getattr(self.backend, "session", None)

    # ---- stream pipeline (corner == "local", HIP sessions): one launch per frame on every rank, nothing between ranks
    def stream_begin(self):
        assert self.corner == "local", "the sharded stream pipeline needs the corner flows computed locally"
        self._pending = None
        self.session.stream_begin()

    def stream_submit(self, frame) -> int:
        """Next frame of the stream; returns the pair whose flow (this rank's rows) the launch writes, or -1."""
        return self.session.stream_submit(frame)

    def stream_submit_frames(self, frames) -> int:
        """Several consecutive frames in one call (engine.FrameGroup or a sequence of device tensors)."""
        return self.session.stream_submit_frames(frames)

    def stream_drain(self) -> int:
        if getattr(self, "_pending", None) is not None:   # a tick whose exchange overlapped the launch before it: submit it first
            done = self._submit_pending()
            if done >= 1:
                return done
        return self.session.stream_drain()

    # ---- frames that ARRIVE sharded, through the stream pipeline (halo_mode == "stream_exchange") ------------------------------
    # north_star's "RCCL halo exchange" for callers whose ranks only ever receive their own rows of a frame, without giving up
    # the one-launch pipeline: what crosses ranks is LEVEL 0 only -- the halo rows this rank's plan holds beyond its block (the
    # halos of the coarser levels are recomputed from them, as in "recompute"), and the rows / columns of the frame's top-left
    # patch, from which every rank forms the shift vectors itself -- one batched group of sends and receives per TICK of B frames,
    # issued for the next tick while the launch of this one runs.  The session then sees an ordinary frame buffer in which the rows
    # it never reads were never written.
    def _exchange_lists(self):
        """[(peer, row0, row1, cols)] to receive and to send: full-width halo rows and the patch's rows and columns"""
        if getattr(self, "_xlists", None) is not None:
            return self._xlists
        p = self.plan
        pw, ph = p.patch_wh(self.patch_size)
        plans = [ShardPlan(p.width, p.height, p.levels, p.window, r, self.world, p.margin, p.halo_mode, iters=p.iters, warp_margin=p.warp_margin)
                 for r in range(self.world)]

        def wants(me, src):   # pieces of src's own rows that rank `me` needs: [(row0, row1, cols)]
            (b0, b1), (o0, o1) = plans[me].buf[0], plans[src].own[0]
            out = []
            a, b = max(b0, o0), min(b1, o1)
            if a < b:
                out.append((a, b, p.width))
            for a, b in ((max(0, o0), min(ph, o1, b0)), (max(o0, b1, 0), min(ph, o1))):   # patch rows outside what `me` holds anyway
                if a < b:
                    out.append((a, b, pw))
            return out
        recv = [(r,) + piece for r in range(self.world) if r != self.rank for piece in wants(self.rank, r)]
        send = [(r,) + piece for r in range(self.world) if r != self.rank for piece in wants(r, self.rank)]
        self._xlists = (recv, send)
        return self._xlists

    def assemble_frames(self, own_rows, buffers):
        """own_rows: this rank's rows [own0, own1) of the n frames of a tick -- a sequence of [rows, width] tensors or one stacked
        [n, rows, width] tensor; buffers: n frame-shaped tensors [height, >= width] (a sequence, or one stacked [n, height, width]
        tensor) that receive them, the halo rows and the patch.  ONE message per peer and direction carries the pieces of all n
        frames (halo rows, patch rows / columns), all of them in one batched group of P2P operations (RCCL over xGMI on GPUs, gloo
        in the CPU tests).  Rows the plan does not hold are left untouched."""
        import torch
        import torch.distributed as dist

        p = self.plan
        o0, o1 = p.own[0]
        stack = own_rows if hasattr(own_rows, "dim") and own_rows.dim() == 3 else torch.stack(list(own_rows))
        n = stack.shape[0]
        assert tuple(stack.shape[1:]) == (o1 - o0, p.width), (tuple(stack.shape), (o1 - o0, p.width))
        stacked_out = hasattr(buffers, "dim") and buffers.dim() == 3

        def put(a, b, cols, block):   # block [n, b - a, cols] -> rows [a, b), columns [0, cols) of every buffer
            if stacked_out:
                buffers[:, a:b, :cols] = block   # one copy for all n frames
            else:
                for i, buf in enumerate(buffers):
                    buf[a:b, :cols] = block[i]

        put(o0, o1, p.width, stack)
        if self.world == 1:
            return
        recv, send = self._exchange_lists()
        ops, inbox = [], []
        packed = {}   # (the rank that owns the patch sends the same piece to every other rank: packed once)

        def piece(a, b, cols):
            if (a, b, cols) not in packed:
                packed[(a, b, cols)] = stack[:, a - o0: b - o0, :cols].reshape(-1).contiguous()
            return packed[(a, b, cols)]
        for peer in sorted({x[0] for x in send}):
            parts = [piece(a, b, cols) for pr, a, b, cols in send if pr == peer]
            ops.append(dist.P2POp(dist.isend, parts[0] if len(parts) == 1 else torch.cat(parts), peer))
        for peer in sorted({x[0] for x in recv}):
            pieces = [(a, b, cols) for pr, a, b, cols in recv if pr == peer]
            t = stack.new_empty((n * sum((b - a) * cols for a, b, cols in pieces),))
            ops.append(dist.P2POp(dist.irecv, t, peer))
            inbox.append((pieces, t))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for pieces, t in inbox:
            off = 0
            for a, b, cols in pieces:
                cnt = n * (b - a) * cols
                put(a, b, cols, t[off: off + cnt].view(n, b - a, cols))
                off += cnt

    def _frame_buffers(self, n, like):
        """n frame-shaped buffers for the assembled frames of a tick, as one stacked [n, height, width] tensor out of a ring of
        groups (borrowed frames stay in use for three ticks: five groups)"""
        key = int(n)
        if not hasattr(self, "_rings"):
            self._rings = {}
        if key not in self._rings:
            self._rings[key] = [like.new_zeros((5, key, self.plan.height, self.plan.width)), 0, {}]
        ring = self._rings[key]
        g = ring[1]
        ring[1] = (g + 1) % ring[0].shape[0]
        return ring[0][g], g, ring[2]

    def stream_submit_own_rows(self, own_rows, overlap=None) -> int:
        """A tick's frames as this rank's own rows only ([n, rows, width] or a sequence): assemble (one exchange) and hand the
        buffers to the stream pipeline in one call.

        overlap (default: on for device tensors when there is something to exchange): the exchange of THIS tick is enqueued on a
        side stream and the launch of the tick BEFORE -- whose exchange has been running since the previous call -- on the caller's
        stream, so that RCCL's sends and receives and the packing copies run underneath a tick's launches instead of between them.
        One more tick of latency: the value returned is that of the tick submitted now (the previous call's frames), and
        stream_drain() first flushes the tick still waiting.  Ordering: the side stream waits for everything enqueued on the
        caller's stream so far -- the rows handed in, and the launches that last read the ring group this tick's buffers come
        from (five groups, a borrowed frame is read for three ticks) --; the caller's stream waits for the exchange's event before
        the launch that reads the assembled buffers."""
        from . import engine

        first = own_rows[0]
        bufs, g, groups = self._frame_buffers(len(own_rows), first)
        if g not in groups:   # (the argument block of a ring group, packed once)
            groups[g] = engine.FrameGroup([bufs[i] for i in range(bufs.shape[0])])
        if overlap is None:
            import os

            env = os.environ.get("OFX_SHARD_OVERLAP")   # (rehearsals on one rank: 1 forces the side stream on, 0 off)
            overlap = (env == "1") if env in ("0", "1") else (bool(getattr(first, "is_cuda", False)) and self.world > 1)
        if not overlap:
            done = self._submit_pending()
            self.assemble_frames(own_rows, bufs)
            return max(done, self.session.stream_submit_frames(groups[g]))
        import torch

        cur = torch.cuda.current_stream()
        if getattr(self, "_xstream", None) is None:
            self._xstream = torch.cuda.Stream()
        x = self._xstream
        x.wait_stream(cur)
        with torch.cuda.stream(x):
            self.assemble_frames(own_rows, bufs)
            ev = torch.cuda.Event()
            ev.record(x)
        if hasattr(own_rows, "record_stream"):
            own_rows.record_stream(x)   # (the caller may drop the tensor right away: its memory is in use on the side stream)
        else:
            for t in own_rows:
                t.record_stream(x)
        done = self._submit_pending()
        self._pending = (groups[g], ev)
        return done

    def _submit_pending(self) -> int:
        """the tick whose exchange was started by the previous stream_submit_own_rows(overlap=True), if any"""
        pend = getattr(self, "_pending", None)
        if pend is None:
            return -1
        import torch

        self._pending = None
        grp, ev = pend
        torch.cuda.current_stream().wait_event(ev)
        return self.session.stream_submit_frames(grp)

    def push_own_rows(self, rows):
        """pair-at-a-time form (the CPU stand-in of the tests): priming with this rank's own rows of the first frame"""
        bufs, _, _ = self._frame_buffers(1, rows)
        self.assemble_frames([rows], bufs)
        self.push_frame(bufs[0])

    def step_own_rows(self, rows):
        bufs, _, _ = self._frame_buffers(1, rows)
        self.assemble_frames([rows], bufs)
        self.step(bufs[0])

    def push_frame(self, frame):
        """Make `frame` the previous frame (priming, main.cu:203-209)."""
        b = self.backend
        b.load_frame(frame)
        self._build_pyramid()
        b.swap()

    # ---- halo exchange (halo_mode == "exchange") -------------------------------------------------------------------
    def _build_pyramid(self):
        b = self.backend
        if self.plan.halo_mode != "exchange":
            b.build_pyramid()
            return
        # only this rank's own rows of the frame count (the rest of what load_frame brought in is overwritten here):
        # level by level, own rows -> neighbours' halos, then the next level's own rows from them
        self.exchange_halos(0)
        for k in range(1, self.plan.levels):
            b.downsample_level(k)
            self.exchange_halos(k)

    def exchange_halos(self, level: int):
        """Fill the halo rows of `level`'s next plane from the neighbouring ranks' own rows (one batched send/recv pair
        per neighbour: RCCL over xGMI on GPUs, gloo in the CPU tests)."""
        import torch.distributed as dist

        if self.world == 1:
            return
        p = self.plan
        plane, base = self.backend.next_plane(level)
        w = p.width >> level
        (o0, o1), (b0, b1) = p.own[level], p.buf[level]
        ops, keep = [], []
        if self.rank > 0:  # upper neighbour: it needs my first rows, I need its last ones
            up = ShardPlan(p.width, p.height, p.levels, p.window, self.rank - 1, self.world, p.margin, p.halo_mode)
            n_send = up.buf[level][1] - up.own[level][1]      # rows it holds below its block = my rows [o0, o0 + n_send)
            n_recv = o0 - b0
            snd = plane[o0 - base: o0 - base + n_send, :w].contiguous()
            rcv = plane.new_empty((n_recv, w))
            ops += [dist.P2POp(dist.isend, snd, self.rank - 1), dist.P2POp(dist.irecv, rcv, self.rank - 1)]
            keep.append((rcv, b0, n_recv))
        if self.rank + 1 < self.world:
            dn = ShardPlan(p.width, p.height, p.levels, p.window, self.rank + 1, self.world, p.margin, p.halo_mode)
            n_send = dn.own[level][0] - dn.buf[level][0]      # rows it holds above its block = my rows [o1 - n_send, o1)
            n_recv = b1 - o1
            snd = plane[o1 - n_send - base: o1 - base, :w].contiguous()
            rcv = plane.new_empty((n_recv, w))
            ops += [dist.P2POp(dist.isend, snd, self.rank + 1), dist.P2POp(dist.irecv, rcv, self.rank + 1)]
            keep.append((rcv, o1, n_recv))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for rcv, row, n in keep:
            plane[row - base: row - base + n, :w] = rcv

    def step(self, frame, check_margin: bool = False):
        """Flow between the previous frame and `frame`; afterwards `frame` is the previous frame."""
        import torch.distributed as dist

        b = self.backend
        if self.corner == "local":
            # pair-at-a-time form of the local mode (the CPU stand-in of the tests; HIP sessions use stream_submit)
            b.load_frame(frame)
            b.build_pyramid()
            b.corner_flows_local()
            b.run_levels()
            b.swap()
            return
        if self.pipelined and not check_margin and hasattr(b, "pipelined_step") and self.plan.halo_mode != "exchange":
            b.pipelined_step(frame, self.rank, self.world)
            return
        b.load_frame(frame)
        self._build_pyramid()
        if self.rank == 0:
            b.corner_flows()  # needs only rank 0's own corner of every level
        if self.world > 1:
            stage = getattr(b, "uv_stage", None)
            if stage is None:
                dist.broadcast(b.uv_all, src=0)
            else:
                if self.rank == 0:
                    stage.copy_(b.uv_all)
                dist.broadcast(stage, src=0)
                b.uv_all.copy_(stage)
        if check_margin:
            self.assert_margin()
        b.run_levels()
        b.swap()

    def corner_status(self) -> int:
        """Device status word of this rank's session, read and cleared (synchronises): 0 = every pair so far is exactly
        the unsharded result; bit k = level k's shift left the top-left patch (local corner mode), bit 8 + k = level k's
        vertical shift reached image rows beyond this shard's halo (any mode; the `margin` rows of the plan).  bench.py
        and long-running drivers check it after a batch of pairs instead of syncing per pair (assert_margin)."""
        return self.session.corner_status() if self.session is not None else 0

    def assert_margin(self):
        """The shift is exact while every level's vertical shift stays within the halo margin (host sync: tests only)."""
        import math

        uv = self.backend.uv_all.detach().cpu().tolist()
        halo = self.plan.window // 2 + 1
        for k in range(self.plan.levels - 1):
            v, hk = uv[2 * k + 1], self.plan.height >> k
            if not math.isfinite(v) or abs(v) <= self.plan.margin:
                continue  # non-finite: every target is outside the image, nothing is fetched (OptFlowCPU.cpp:270)
            y0 = max(0, self.plan.own[k][0] - halo)
            y1 = min(hk, self.plan.own[k][1] + halo)
            lo, hi = y0 + v, (y1 - 1) + v  # targets of the rows this rank shifts
            if hi > -1 and lo < hk:        # some target lands inside the image: its row must be in the buffer
                raise RuntimeError(f"level {k}: vertical shift {v:.2f} exceeds the halo margin {self.plan.margin}; "
                                   "re-create the plan with a larger margin")

    def gather_flow(self, level: int):
        """Full (h_k, w_k, 2) flow of a level on every rank (tests / host consumers)."""
        import torch
        import torch.distributed as dist

        mine = self.backend.flow(level).contiguous()
        if self.world == 1:
            return mine
        w = self.plan.width >> level
        rows = [ShardPlan(self.plan.width, self.plan.height, self.plan.levels, self.plan.window, r, self.world,
                          self.plan.margin).own[level] for r in range(self.world)]
        most = max(b - a for a, b in rows)
        padded = torch.zeros((most, w, 2), dtype=mine.dtype, device=mine.device)  # equal-sized pieces for all_gather
        padded[: mine.shape[0]] = mine
        parts = [torch.empty_like(padded) for _ in rows]
        dist.all_gather(parts, padded)
        return torch.cat([p[: b - a] for p, (a, b) in zip(parts, rows)], dim=0)
