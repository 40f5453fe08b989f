"""Host-side mirror of the engine's C ABI (include/ofx.h) for Python callers.

PyTorch is used only as plumbing: device memory (``torch.empty(..., device='cuda')``), streams and, in
``parallel.py``, ``torch.distributed``.  All arithmetic runs in libofx_hip.so; there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import lib as _lib
from .lib import Geom, MODES, OfxError, Params, check

_vp = C.c_void_p


def _stream_ptr(stream=None) -> Optional[int]:
    """hipStream_t of a torch stream (None -> torch's current stream)."""
    import torch

    st = stream if stream is not None else torch.cuda.current_stream()
    return st.cuda_stream or None


def pitch_for(w: int) -> int:
    return (w + 63) // 64 * 64


class DeviceView:
    """Zero-copy torch view of a device range owned by the C session (via __cuda_array_interface__)."""

    def __init__(self, ptr: int, shape, typestr: str, strides=None):
        self.__cuda_array_interface__ = {"data": (int(ptr), False), "shape": tuple(shape), "typestr": typestr,
                                         "strides": strides, "version": 2}

    def tensor(self):
        import torch

        return torch.as_tensor(self, device="cuda")


class FrameGroup:
    """Consecutive frames of a stream as one argument block for Session.stream_submit_frames (keeps the tensors alive)."""

    def __init__(self, tensors):
        self.tensors = list(tensors)
        self.n = len(self.tensors)
        assert self.n >= 1 and all(t.is_cuda and t.dtype.itemsize == 1 and t.stride(0) == self.tensors[0].stride(0) for t in self.tensors)
        self.pitch = int(self.tensors[0].stride(0))
        self.ptrs = (_vp * self.n)(*[t.data_ptr() for t in self.tensors])


def suggest_stream_batch(width: int, height: int, levels: int, shard=None, borrow_frames: bool = False, two_stage: bool = False) -> int:
    """Frames per launch (ofx_params.stream_batch) for a throughput-bound stream on MI355X: the largest B in {16, 8, 4} the
    launch can carry (OFX_MAX_LK_ITEMS = 80 (pair, level) items) whose cyclic working set stays inside the 256 MB Infinity
    Cache, else 2.  The working set is what lives between a frame's arrival and its last use, d = 3 ticks later (2 with
    two_stage = ofx_params.stream_two_stage): the session's dB+2 image sets (the whole pyramid, or levels >= 1 only with
    borrow_frames) plus the caller's ring of frames (borrow_frames needs at least dB+1 buffers -- include/ofx.h states the
    lifetime rule: frame f's buffer is read until the launch enqueued by the submit of frame f+dB has run --; the estimate
    assumes the ring bench.py uses, dB+4 rounded up to a multiple of four).  Measured: the fused level
    kernel runs ~25 % slower when its image rows come from HBM instead of that cache -- 4K, copied frames: 2 / 4 frames per
    launch = 222k / 197k Mpix/s; borrowed frames in three stages: 4 / 8 = 251-264k / 219k, in two stages: 8 = 274-280k; 1080p
    takes 16 (8 / 16 frames per launch = 227k-247k / 258k-267k), the ranks of a sharded pair 8 (sixteen measured no better
    there: tools/shard_sim.py)."""
    rows = [(height >> k) if shard is None else (shard.buf[k][1] - shard.buf[k][0]) for k in range(levels)]
    level_bytes = [(width >> k) * rows[k] for k in range(levels)]
    depth = 2 if two_stage else 3   # ticks a frame stays in use (ofx_params.stream_two_stage)
    for b in ((16, 8, 4) if shard is None else (8, 4)):  # (a rank of a sharded pair gains nothing from sixteen: measured)
        ring = (depth * max(b, 4) + 4 + 3) // 4 * 4
        working_set = (depth * b + 2) * sum(level_bytes[1 if borrow_frames else 0:]) + ring * level_bytes[0]
        if b * levels <= 80 and working_set <= 230e6:
            return b
    return 2


class Session:
    """Device-resident frame loop (main.cu:192-272): ofx_session_* behind a small object."""

    def __init__(self, width: int, height: int, levels: int, window: int, mode: str = "lk_float", device: int = 0,
                 shard=None, iters: int = 1, local_corner: bool = False, patch_size: int = 0, stream_batch: int = 1, borrow_frames: bool = False,
                 min_det: float = 0.0, two_stage: bool = False, strict: bool = True, frames_partial: bool = False, deep_fetch: int = 0):
        """strict: stream_drain() raises OfxError when the pipeline has drained and the session's status word is not 0 -- a pair
        whose result is NOT the reference's (a corner shift that left the patch of a session that cannot repair it, a shift or
        warp beyond a shard's halo; include/ofx.h, ofx_session_corner_status).  strict=False: poll corner_status() yourself."""
        self.L = _lib.load()
        self.strict = bool(strict)
        self._status = 0
        self.width, self.height, self.levels, self.window, self.mode = width, height, levels, window, mode
        p = Params()
        p.width, p.height, p.levels, p.window, p.mode, p.device = width, height, levels, window, MODES[mode], device
        p.iters = iters
        p.local_corner, p.patch_size = int(bool(local_corner)), int(patch_size)
        p.stream_batch = int(stream_batch)
        p.borrow_frames = int(bool(borrow_frames))
        p.min_det = float(min_det)
        p.stream_two_stage = int(bool(two_stage))
        p.frames_partial = int(bool(frames_partial))   # (ofx_params.frames_partial: only the plan's rows + the patch were ever written)
        # ofx_params.deep_fetch: +1 = the frames handed to the stream pipeline are cold (last touched more than an Infinity Cache of
        # traffic ago), -1 = warm, 0 = decide by level size.  Speed only; the bits are the same.
        p.deep_fetch = int(deep_fetch)
        self.shard = shard
        if shard is not None:
            p.sharded = 1
            for k in range(levels):
                p.own_y0[k], p.own_y1[k] = shard.own[k]
                p.buf_y0[k], p.buf_y1[k] = shard.buf[k]
                p.comp_y0[k], p.comp_y1[k] = shard.comp[k]
        self._h = _vp()
        check(self.L.ofx_session_create(C.byref(p), C.byref(self._h)), "ofx_session_create")
        self._keep = []

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            check(self.L.ofx_session_destroy(self._h), "ofx_session_destroy")
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- frames
    def set_frame_host(self, gray1: np.ndarray, stream=None):
        gray1 = np.ascontiguousarray(gray1, dtype=np.uint8)
        assert gray1.shape == (self.height, self.width), gray1.shape
        self._keep = [gray1]
        check(self.L.ofx_session_set_frame_host(self._h, gray1.ctypes.data, _stream_ptr(stream)), "set_frame_host")

    def set_frame_host_3ch(self, img3: np.ndarray, stream=None):
        img3 = np.ascontiguousarray(img3, dtype=np.uint8)
        assert img3.shape == (self.height, self.width, 3), img3.shape
        self._keep = [img3]
        check(self.L.ofx_session_set_frame_host_3ch(self._h, img3.ctypes.data, _stream_ptr(stream)), "set_frame_host_3ch")

    def set_frame_device(self, t, stream=None):
        """t: torch uint8 CUDA tensor (height, width), row stride = t.stride(0)."""
        assert t.is_cuda and t.dtype.itemsize == 1 and tuple(t.shape) == (self.height, self.width)
        check(self.L.ofx_session_set_frame_device(self._h, t.data_ptr(), int(t.stride(0)), _stream_ptr(stream)),
              "set_frame_device")

    # ---- steps
    def build_pyramid(self, stream=None):
        check(self.L.ofx_session_build_pyramid(self._h, _stream_ptr(stream)), "build_pyramid")

    def downsample_level(self, level: int, stream=None):
        check(self.L.ofx_session_downsample_level(self._h, level, _stream_ptr(stream)), "downsample_level")

    def run_flow(self, stream=None):
        check(self.L.ofx_session_run_flow(self._h, _stream_ptr(stream)), "run_flow")

    def run_flow_sequential(self, stream=None):
        check(self.L.ofx_session_run_flow_sequential(self._h, _stream_ptr(stream)), "run_flow_sequential")

    def corner_flows(self, stream=None):
        check(self.L.ofx_session_corner_flows(self._h, _stream_ptr(stream)), "corner_flows")

    def run_levels(self, stream=None):
        check(self.L.ofx_session_run_levels(self._h, _stream_ptr(stream)), "run_levels")

    def compute_uv(self, level: int, stream=None):
        check(self.L.ofx_session_compute_uv(self._h, level, _stream_ptr(stream)), "compute_uv")

    def run_level(self, level: int, stream=None):
        check(self.L.ofx_session_run_level(self._h, level, _stream_ptr(stream)), "run_level")

    def swap(self):
        check(self.L.ofx_session_swap(self._h), "swap")

    # ---- pipelined pair: staging on the session's aux stream under the previous pair's LK launch
    def submit_device(self, t, stream=None):
        assert t.is_cuda and t.dtype.itemsize == 1 and tuple(t.shape) == (self.height, self.width)
        check(self.L.ofx_session_submit_device(self._h, t.data_ptr(), int(t.stride(0)), _stream_ptr(stream)), "submit_device")

    def stage_frame(self, t, aux=None):
        check(self.L.ofx_session_stage_frame(self._h, t.data_ptr(), int(t.stride(0)), aux), "stage_frame")

    def stage_shift(self, aux=None):
        check(self.L.ofx_session_stage_shift(self._h, aux), "stage_shift")

    def solve_staged(self, stream=None):
        check(self.L.ofx_session_solve_staged(self._h, _stream_ptr(stream)), "solve_staged")

    def aux_stream_ptr(self) -> int:
        p = _vp()
        check(self.L.ofx_session_aux_stream(self._h, C.byref(p)), "aux_stream")
        return p.value

    # ---- stream pipeline: one launch per frame, flow of pair p ready after frame p+3
    def corner_status(self, stream=None) -> int:
        """local_corner sessions: OR of the "shift left the patch at level k" bits since the last call (0 = all exact)."""
        st = C.c_int(0)
        check(self.L.ofx_session_corner_status(self._h, C.byref(st), _stream_ptr(stream)), "corner_status")
        word, self._status = int(st.value) | self._status, 0
        return word

    STATUS_REPAIRED = 1 << 24   # OFX_STATUS_REPAIRED

    def pair_status(self, pair: int, stream=None) -> int:
        """Status word of ONE pair of the stream pipeline (ofx_session_pair_status): the error bits that pair raised, plus
        STATUS_REPAIRED when its shifted corner left the top-left patch and was read through a relocated one (exact all the same)."""
        st = C.c_int(0)
        check(self.L.ofx_session_pair_status(self._h, pair, C.byref(st), _stream_ptr(stream)), "pair_status")
        return int(st.value)

    def stream_begin(self):
        check(self.L.ofx_session_stream_begin(self._h), "stream_begin")

    def stream_submit(self, t, stream=None) -> int:
        assert t.is_cuda and t.dtype.itemsize == 1 and tuple(t.shape) == (self.height, self.width)
        done = C.c_int(-1)
        check(self.L.ofx_session_stream_submit(self._h, t.data_ptr(), int(t.stride(0)), _stream_ptr(stream), C.byref(done)),
              "stream_submit")
        return done.value

    def stream_submit_frames(self, frames, stream=None) -> int:
        """Several consecutive frames in one call (ofx_session_stream_submit_frames): one FFI crossing per group instead of
        one per frame.  `frames`: a sequence of device tensors of equal pitch, or a FrameGroup packed once for a ring that is
        reused.  Returns the highest pair complete after the call, or -1."""
        g = frames if isinstance(frames, FrameGroup) else FrameGroup(frames)
        done = C.c_int(-1)
        check(self.L.ofx_session_stream_submit_frames(self._h, g.ptrs, None, g.pitch, g.n, _stream_ptr(stream), C.byref(done)),
              "stream_submit_frames")
        return done.value

    def stream_drain(self, stream=None) -> int:
        done = C.c_int(-1)
        check(self.L.ofx_session_stream_drain(self._h, _stream_ptr(stream), C.byref(done)), "stream_drain")
        if done.value == -2 and self.strict:
            # the pipeline is empty: every pair it produced must have been the reference's (one synchronising read per drained
            # stream, not per frame).  The word is kept for corner_status() when the caller wants to look at it.
            st = C.c_int(0)
            check(self.L.ofx_session_corner_status(self._h, C.byref(st), _stream_ptr(stream)), "corner_status")
            self._status |= int(st.value)
            if self._status:
                word, self._status = self._status, 0
                raise _lib.OfxError(f"stream pipeline: status word {word:#x} -- bit k: level k's corner shift left the patch and could not be "
                                    "repaired; bit 8+k / 16+k: a shift / warp reached beyond this shard's halo: those pairs are not the "
                                    "reference's result (include/ofx.h, ofx_session_corner_status)")
        return done.value

    def push_frame_host(self, gray1: np.ndarray, stream=None):
        """Load a frame, build its pyramid and make it the previous frame (priming step of main.cu:203-209)."""
        self.set_frame_host(gray1, stream)
        self.build_pyramid(stream)
        self.swap()

    # ---- timing of the level-0 fused kernel (HIP events on the launch stream)
    def timing(self, max_launches: int):
        check(self.L.ofx_session_timing(self._h, max_launches), "session_timing")

    def timing_read(self):
        avg, mn, n = C.c_double(), C.c_double(), C.c_int()
        check(self.L.ofx_session_timing_read(self._h, C.byref(avg), C.byref(mn), C.byref(n)), "session_timing_read")
        return avg.value, mn.value, n.value

    TIME_KINDS = {"lk": 0, "lk_acc": 1, "warp": 2, "stream": 3, "shift": 4, "corner": 5, "pyramid": 6, "lk_acc_warp": 7}   # OFX_TIME_*

    def timing_read_kind(self, kind: str):
        """(average us, minimum us, launches) of one kind of launch; call before timing_read, which re-arms."""
        avg, mn, n = C.c_double(), C.c_double(), C.c_int()
        check(self.L.ofx_session_timing_read_kind(self._h, self.TIME_KINDS[kind], C.byref(avg), C.byref(mn), C.byref(n)),
              "session_timing_read_kind")
        return avg.value, mn.value, n.value

    # ---- buffers
    def plane(self, which: int, level: int):
        """(torch uint8 view [rows, pitch], Geom) of plane 0=prev 1=next 2=shifted."""
        ptr, g = _vp(), Geom()
        check(self.L.ofx_session_plane(self._h, which, level, C.byref(ptr), C.byref(g)), "session_plane")
        return DeviceView(ptr.value, (g.rows, g.pitch), "|u1").tensor(), g

    def flow(self, level: int):
        """torch float32 view [own_rows, w, 2] of the level's flow, and the global index of its first row."""
        ptr, r0, rows = _vp(), C.c_int(), C.c_int()
        check(self.L.ofx_session_flow(self._h, level, C.byref(ptr), C.byref(r0), C.byref(rows)), "session_flow")
        w = self.width >> level
        return DeviceView(ptr.value, (rows.value, w, 2), "<f4").tensor(), r0.value

    def flow_of(self, pair: int, level: int):
        """As flow(), for one of the newest completed pairs of the stream pipeline (two of them with stream_batch = 2)."""
        ptr, r0, rows = _vp(), C.c_int(), C.c_int()
        check(self.L.ofx_session_flow_of(self._h, pair, level, C.byref(ptr), C.byref(r0), C.byref(rows)), "session_flow_of")
        w = self.width >> level
        return DeviceView(ptr.value, (rows.value, w, 2), "<f4").tensor(), r0.value

    def uv(self, level: int):
        """Shift vector of `level` for the pair in progress (the slot alternates per pair: query after every swap)."""
        ptr = _vp()
        check(self.L.ofx_session_shift_uv(self._h, level, C.byref(ptr)), "session_shift_uv")
        return DeviceView(ptr.value, (2,), "<f4").tensor()

    def flow_host(self, level: int, stream=None) -> np.ndarray:
        ptr, r0, rows = _vp(), C.c_int(), C.c_int()
        check(self.L.ofx_session_flow(self._h, level, C.byref(ptr), C.byref(r0), C.byref(rows)), "session_flow")
        out = np.empty((rows.value, self.width >> level, 2), np.float32)
        check(self.L.ofx_session_get_flow_host(self._h, level, out.ctypes.data, _stream_ptr(stream)), "get_flow_host")
        return out


# ---- stateless device-pointer calls on torch tensors -------------------------------------------------------------

def _u8_plane(arr: np.ndarray):
    """Upload an (h, w) u8 array into a pitched CUDA tensor; returns (tensor[h, pitch], pitch)."""
    import torch

    h, w = arr.shape
    pitch = pitch_for(w)
    t = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda")
    t[:, :w] = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
    return t, pitch


def lk_level(prev1: np.ndarray, next1: np.ndarray, window: int, mode: str, want_sums: bool = False,
             rows: Optional[Sequence[int]] = None, buf_rows: Optional[Sequence[int]] = None):
    """One fused LK level on host arrays (upload, ofx_lk_level[_sums], download).

    rows=(y0,y1) restricts the produced rows; buf_rows=(r0,r1) uploads only those rows (shard emulation)."""
    import torch

    L = _lib.load()
    h, w = prev1.shape
    r0, r1 = buf_rows if buf_rows is not None else (0, h)
    y0, y1 = rows if rows is not None else (0, h)
    tp, pitch = _u8_plane(prev1[r0:r1])
    tn, _ = _u8_plane(next1[r0:r1])
    g = Geom(w, h, pitch, r0, r1 - r0, y0, y1)
    st = _stream_ptr()
    if want_sums:
        out = torch.zeros((5, y1 - y0, w), dtype=torch.int32, device="cuda")
        check(L.ofx_lk_level_sums(tp.data_ptr(), tn.data_ptr(), C.byref(g), window, MODES[mode], out.data_ptr(), y0, st),
              "ofx_lk_level_sums")
    else:
        out = torch.zeros((y1 - y0, w, 2), dtype=torch.float32, device="cuda")
        check(L.ofx_lk_level(tp.data_ptr(), tn.data_ptr(), C.byref(g), window, MODES[mode], out.data_ptr(), y0, st),
              "ofx_lk_level")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def downsample_1ch(src1: np.ndarray) -> np.ndarray:
    import torch

    L = _lib.load()
    sh, sw = src1.shape
    h, w = sh >> 1, sw >> 1
    ts, sp = _u8_plane(src1)
    dp = pitch_for(w)
    td = torch.zeros((h, dp), dtype=torch.uint8, device="cuda")
    g = Geom.full(w, h, dp)
    check(L.ofx_downsample_1ch(ts.data_ptr(), sp, 0, sh, td.data_ptr(), C.byref(g), _stream_ptr()), "ofx_downsample_1ch")
    torch.cuda.synchronize()
    return td[:, :w].cpu().numpy()


def shift_1ch(src1: np.ndarray, uv) -> np.ndarray:
    import torch

    L = _lib.load()
    h, w = src1.shape
    ts, pitch = _u8_plane(src1)
    td = torch.zeros_like(ts)
    tuv = torch.tensor(list(uv), dtype=torch.float32, device="cuda")
    g = Geom.full(w, h, pitch)
    check(L.ofx_shift_1ch(ts.data_ptr(), td.data_ptr(), C.byref(g), tuv.data_ptr(), _stream_ptr()), "ofx_shift_1ch")
    torch.cuda.synchronize()
    return td[:, :w].cpu().numpy()


def shift_vector(flow_levels: List[Optional[np.ndarray]], level: int, max_level: int) -> np.ndarray:
    import torch

    L = _lib.load()
    keep, ptrs = [], (_vp * _lib.OFX_MAX_LEVELS)()
    for k in range(level + 1, max_level):
        t = torch.from_numpy(np.ascontiguousarray(flow_levels[k], dtype=np.float32)).cuda()
        keep.append(t)
        ptrs[k] = t.data_ptr()
    out = torch.zeros(2, dtype=torch.float32, device="cuda")
    check(L.ofx_shift_vector(ptrs, level, max_level, out.data_ptr(), _stream_ptr()), "ofx_shift_vector")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def compose_flow(flow_levels: List[np.ndarray], levels: int, level: int) -> np.ndarray:
    import torch

    L = _lib.load()
    keep, ptrs = [], (_vp * _lib.OFX_MAX_LEVELS)()
    for k in range(level, levels):
        t = torch.from_numpy(np.ascontiguousarray(flow_levels[k], dtype=np.float32)).cuda()
        keep.append(t)
        ptrs[k] = t.data_ptr()
    h, w, _ = flow_levels[level].shape
    out = torch.zeros((h, w, 2), dtype=torch.float32, device="cuda")
    check(L.ofx_compose_flow(ptrs, w, h, levels, level, out.data_ptr(), _stream_ptr()), "ofx_compose_flow")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def warp_u8(src1: np.ndarray, flow: np.ndarray, scale: float) -> np.ndarray:
    import torch
    from .lib import Geom

    class WarpDesc(C.Structure):
        _fields_ = [("d_src", _vp), ("d_dst", _vp), ("geom", Geom), ("d_flow", _vp), ("flow_row0", C.c_int), ("scale", C.c_float),
                    ("d_status", _vp), ("status_bit", C.c_int)]

    L = _lib.load()
    h, w = src1.shape
    ts, pitch = _u8_plane(src1)
    td = torch.zeros_like(ts)
    tf = torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32)).cuda()
    d = WarpDesc(ts.data_ptr(), td.data_ptr(), Geom.full(w, h, pitch), tf.data_ptr(), 0, scale, None, 0)
    check(L.ofx_warp_levels(C.byref(d), 1, _stream_ptr()), "ofx_warp_levels")
    torch.cuda.synchronize()
    return td[:, :w].cpu().numpy()


def flow_pair(prev1: np.ndarray, next1: np.ndarray, levels: int, window: int, mode: str, iters: int = 1, min_det: float = 0.0) -> List[np.ndarray]:
    """Whole pair through a Session: returns the flow pyramid as host arrays."""
    import torch

    h, w = prev1.shape
    s = Session(w, h, levels, window, mode, iters=iters, min_det=min_det)
    try:
        s.push_frame_host(prev1)
        s.set_frame_host(next1)
        s.build_pyramid()
        s.run_flow()
        torch.cuda.synchronize()
        return [s.flow_host(k) for k in range(levels)]
    finally:
        s.close()
