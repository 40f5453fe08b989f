"""ctypes binding of libofx_hip.so (the C ABI declared in include/ofx.h).

There is no CPU fallback: if the library is missing or fails to load, importing the engine raises.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, os.environ.get("OFX_LIB", "libofx_hip.so"))  # OFX_LIB: A/B an alternative build

OFX_MAX_LEVELS = 12
MODE_COMPAT_CPU = 0
MODE_LK_FLOAT = 1
MODE_LK_FLOAT_FAST = 2   # lk_float with the <= 1 ulp solve (include/ofx.h)
MODES = {"compat_cpu": MODE_COMPAT_CPU, "lk_float": MODE_LK_FLOAT, "lk_float_fast": MODE_LK_FLOAT_FAST}
SOLVE_F64, SOLVE_INLINE_CPU, SOLVE_F32 = 0, 1, 2


class OfxError(RuntimeError):
    pass


class Geom(C.Structure):
    _fields_ = [("w", C.c_int), ("h", C.c_int), ("pitch", C.c_int), ("row0", C.c_int), ("rows", C.c_int),
                ("out_y0", C.c_int), ("out_y1", C.c_int)]

    @classmethod
    def full(cls, w, h, pitch):
        return cls(w, h, pitch, 0, h, 0, h)


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("levels", C.c_int), ("window", C.c_int), ("mode", C.c_int),
                ("device", C.c_int), ("sharded", C.c_int),
                ("own_y0", C.c_int * OFX_MAX_LEVELS), ("own_y1", C.c_int * OFX_MAX_LEVELS),
                ("buf_y0", C.c_int * OFX_MAX_LEVELS), ("buf_y1", C.c_int * OFX_MAX_LEVELS),
                ("comp_y0", C.c_int * OFX_MAX_LEVELS), ("comp_y1", C.c_int * OFX_MAX_LEVELS),
                ("iters", C.c_int), ("local_corner", C.c_int), ("patch_size", C.c_int), ("stream_batch", C.c_int), ("borrow_frames", C.c_int), ("min_det", C.c_float),
                ("stream_two_stage", C.c_int), ("frames_partial", C.c_int), ("deep_fetch", C.c_int)]


_vp = C.c_void_p
_i = C.c_int
_d = C.c_double
_gp = C.POINTER(Geom)

# name -> argtypes; every function returns int unless listed in _RESTYPE
_SIGS = {
    "ofx_abi_version": [],
    "ofx_device_count": [],
    "ofx_lk_level": [_vp, _vp, _gp, _i, _i, _vp, _i, _vp],
    "ofx_lk_levels": [_vp, _i, _i, _i, _vp],
    "ofx_corner_flows": [_vp, _i, _i, _i, _vp, _vp],
    "ofx_shift_levels": [_vp, _i, _vp],
    "ofx_lk_level_sums": [_vp, _vp, _gp, _i, _i, _vp, _i, _vp],
    "ofx_downsample_1ch": [_vp, _i, _i, _i, _vp, _gp, _vp],
    "ofx_pyramid_1ch": [_vp, _i, _i, _i, C.POINTER(_vp), C.POINTER(_i), _i, _vp],
    "ofx_shift_vector": [C.POINTER(_vp), _i, _i, _vp, _vp],
    "ofx_shift_1ch": [_vp, _vp, _gp, _vp, _vp],
    "ofx_warp_levels": [_vp, _i, _vp],
    "ofx_compose_flow": [C.POINTER(_vp), _i, _i, _i, _i, _vp, _vp],
    "ofx_extract_ch0": [_vp, _vp, _i, _i, _i, _vp],
    "ofx_replicate_3ch": [_vp, _i, _vp, _i, _i, _vp],
    "ofx_grayscale_avg_3ch": [_vp, _vp, _i, _i, _vp],
    "ofx_conv_3ch": [_vp, _vp, _i, _i, _vp, _i, _i, _i, _vp],
    "ofx_conv_3ch_1ch_u8": [_vp, _i, _i, _vp, _vp, _i, _i, _vp],
    "ofx_conv_3ch_1ch_f32": [_vp, _i, _i, _vp, _vp, _i, _i, _vp],
    "ofx_downsample_3ch": [_vp, _vp, _i, _i, _vp],
    "ofx_srm_u8": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "ofx_srm_f32": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "ofx_solve_i32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "ofx_solve_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "ofx_generate_gaussian_kernel": [_d, _i, _vp],
    "ofx_bilateral_3ch": [_vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _vp],
    "ofx_bilateral_3ch_fast": [_vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _vp],
    "ofx_bilateral_wrappers_fast": [_i],
    "ofx_sub_u8": [_vp, _vp, C.c_size_t, _vp, _vp],
    "ofx_srm_3ch_u8": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "ofx_downscale_mask_3ch": [_vp, _vp, _i, _i, _vp, _i, _i, _vp],
    "ofx_shift_3ch": [_vp, _vp, _i, _i, _vp, _vp],
    "ofx_session_create": [C.POINTER(Params), C.POINTER(_vp)],
    "ofx_session_destroy": [_vp],
    "ofx_session_set_frame_host": [_vp, _vp, _vp],
    "ofx_session_set_frame_host_3ch": [_vp, _vp, _vp],
    "ofx_session_set_frame_device": [_vp, _vp, _i, _vp],
    "ofx_session_build_pyramid": [_vp, _vp],
    "ofx_session_downsample_level": [_vp, _i, _vp],
    "ofx_session_run_flow": [_vp, _vp],
    "ofx_session_corner_flows": [_vp, _vp],
    "ofx_session_run_levels": [_vp, _vp],
    "ofx_session_run_flow_sequential": [_vp, _vp],
    "ofx_session_compute_uv": [_vp, _i, _vp],
    "ofx_session_run_level": [_vp, _i, _vp],
    "ofx_session_swap": [_vp],
    "ofx_session_stream_begin": [_vp],
    "ofx_session_corner_status": [_vp, C.POINTER(_i), _vp],
    "ofx_session_pair_status": [_vp, _i, C.POINTER(_i), _vp],
    "ofx_stage_threads": [],
    "ofx_debug_stream_trace": [_vp, _i, C.POINTER(_i)],
    "ofx_session_flow_of": [_vp, _i, _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i)],
    "ofx_session_stream_submit": [_vp, _vp, _i, _vp, C.POINTER(_i)],
    "ofx_session_stream_drain": [_vp, _vp, C.POINTER(_i)],
    "ofx_session_stream_submit_frames": [_vp, C.POINTER(_vp), C.POINTER(_i), _i, _i, _vp, C.POINTER(_i)],
    "ofx_stream_launch": [_vp, _i, _i, _vp],
    "ofx_session_submit_device": [_vp, _vp, _i, _vp],
    "ofx_session_stage_frame": [_vp, _vp, _i, _vp],
    "ofx_session_stage_shift": [_vp, _vp],
    "ofx_session_solve_staged": [_vp, _vp],
    "ofx_session_aux_stream": [_vp, C.POINTER(_vp)],
    "ofx_session_plane": [_vp, _i, _i, C.POINTER(_vp), _gp],
    "ofx_session_flow": [_vp, _i, C.POINTER(_vp), C.POINTER(_i), C.POINTER(_i)],
    "ofx_session_shift_uv": [_vp, _i, C.POINTER(_vp)],
    "ofx_session_get_flow_host": [_vp, _i, _vp, _vp],
    "ofx_session_timing": [_vp, _i],
    "ofx_session_timing_read": [_vp, C.POINTER(_d), C.POINTER(_d), C.POINTER(_i)],
    "ofx_session_timing_read_kind": [_vp, _i, C.POINTER(_d), C.POINTER(_d), C.POINTER(_i)],
    "ofx_compose_flow_host": [C.POINTER(_vp), _i, _i, _i, _i, _vp],
    "ofx_calc_opt_flow_host": [_vp, _vp, _i, _i, C.POINTER(_vp), _i, _i, _i, _i],
}
_RESTYPE = {"ofx_generate_gaussian_kernel": None}

EXPORTS = sorted(list(_SIGS) + ["ofx_last_error"])

_lib = None


def load() -> C.CDLL:
    """Load libofx_hip.so once.  Raises if it has not been built (python -m cuda_optical_flow_2_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OfxError(f"{LIB_PATH} is missing: build it with `python -m cuda_optical_flow_2_amd.build` "
                       "(hipcc, gfx950). This engine has no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).  Import
    # torch first so that our library binds to the runtime torch's tensors and streams live in; loading ours first
    # would bring in a second, different runtime (on the GPU boxes of this pool that one does not even see the device).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = _RESTYPE.get(name, C.c_int)
    lib.ofx_last_error.restype = C.c_char_p
    lib.ofx_last_error.argtypes = []
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ofx_last_error()
        raise OfxError(f"{what or 'ofx call'} failed (code {rc}): {msg.decode() if msg else ''}")
