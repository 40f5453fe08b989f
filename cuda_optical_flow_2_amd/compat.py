"""numpy front-end to the reference-shaped C++ surface of libofx_hip.so (namespace gpu, include/OptFlowGpu.cuh).

Functions are bound by their Itanium-mangled names -- the very symbols the reference's main.cu links against
 -- so a test written against this class exercises the drop-in boundary itself.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import lib as _lib

_u8p, _f32p, _i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32)

GPU_SYMBOLS = {
    "grayscale_avg": "_ZN3gpu13grayscale_avgEPKhPhii",
    "conv_3ch_2d": "_ZN3gpu11conv_3ch_2dEPKhPhiiPKfii",
    "conv_3ch_2d_constant": "_ZN3gpu20conv_3ch_2d_constantEPKhPhiiPKfii",
    "conv_3ch_tiled": "_ZN3gpu14conv_3ch_tiledEPKhPhiiPKfii",
    "conv_3ch_1ch_constant": "_ZN3gpu21conv_3ch_1ch_constantEPKhiiPhPKfii",
    "conv_3ch_1ch_tiled": "_ZN3gpu18conv_3ch_1ch_tiledEPKhiiPhPKfii",
    "conv_3ch_1ch_tiled_uchar_float": "_ZN3gpu30conv_3ch_1ch_tiled_uchar_floatEPKhiiPfPKfii",
    "conv_1d_3ch": "_ZN3gpu11conv_1d_3chEPhiiS0_",
    "gauss_pyramid": "_ZN3gpu13gauss_pyramidEPPhiiiPKfii",
    "srm_1ch": "_ZN3gpu7srm_1chEPKhS1_iiiiPi",
    "srm_1ch_float": "_ZN3gpu13srm_1ch_floatEPKfS1_iiiiPf",
    "srm_1ch_tiled": "_ZN3gpu13srm_1ch_tiledEPKhS1_iiiiPi",
    "inverse_matrix": "_ZN3gpu14inverse_matrixEPiS0_S0_S0_S0_PPfiii",
    "inverse_matrix_float": "_ZN3gpu20inverse_matrix_floatEPfS0_S0_S0_S0_PS0_iii",
    "calc_opt_flow": "_ZN3gpu13calc_opt_flowEPKhPhiiPPfii",
    "bilinear_filter": "_ZN3gpu15bilinear_filterEPhS0_S0_iiiidd",
}
CPU_SYMBOLS = {   # the reference's OptFlowCpu.hpp:3-184
    "sub_arr": "_ZN3cpu7sub_arrEPhS0_iS0_",
    "grayscale_avg_cpu": "_ZN3cpu17grayscale_avg_cpuEPKhPhii",
    "conv_3ch": "_ZN3cpu8conv_3chEPKhPKfPhiiii",
    "conv_3ch_to_1ch": "_ZN3cpu15conv_3ch_to_1chEPKhiiPhPKfii",
    "downscale_gaussian": "_ZN3cpu18downscale_gaussianEPhiiS0_PKfii",
    "gauss_pyramid": "_ZN3cpu13gauss_pyramidEPPhiiiPKfii",
    "srm_1ch": "_ZN3cpu7srm_1chEPKhS1_iiiiPi",
    "srm_3ch": "_ZN3cpu7srm_3chEPhS0_iiiiPi",
    "shift_back_pyramid": "_ZN3cpu18shift_back_pyramidEPKhiiiiPPfPh",
    "inverse_matrix": "_ZN3cpu14inverse_matrixEPiS0_S0_S0_S0_PPfiii",
    "calc_optical_flow": "_ZN3cpu17calc_optical_flowEPKhPhiiPPfii",
    "bilinear_filter_3ch": "_ZN3cpu19bilinear_filter_3chEPhS0_S0_iiiidd",
}
UTILS_SYMBOLS = {
    "cleanup_outliers": "_ZN5utils16cleanup_outliersEPhii",
    "upscale_3ch": "_ZN5utils11upscale_3chEPhiiiS0_",
    "upscale_1ch": "_ZN5utils11upscale_1chEPhiiiS0_",
    "generate_gaussian_kernel": "_ZN5utils24generate_gaussian_kernelEdiPd",
}
MASK_SYMBOLS = ["Dx_3x3", "Dx_3x3_t", "Dy_3x3", "Dt_3x3", "Dt_3x3_n", "Dy_DIAGONAL_2x2", "Dy_2x2", "Dz_2x2", "Dx_5x5",
                "GAUS_KERNEL_5x5", "GAUS_KERNEL_3x3"]


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a, t):
    return a.ctypes.data_as(t)


def _ptrs(levels, ctype):
    arr = (C.POINTER(ctype) * len(levels))()
    for i, lv in enumerate(levels):
        arr[i] = lv.ctypes.data_as(C.POINTER(ctype))
    return arr


class GpuCompat:
    def __init__(self):
        self.lib = _lib.load()
        for name in MASK_SYMBOLS:
            n = 25 if "5x5" in name else 9
            setattr(self, name, np.array((C.c_float * n).in_dll(self.lib, name), dtype=np.float32))

    def _f(self, name):
        f = getattr(self.lib, GPU_SYMBOLS[name])
        f.restype = None
        return f

    def _done(self, what):
        rc = self.lib.gpu_compat_last_status()
        if rc != 0:
            raise _lib.OfxError(f"gpu::{what} failed (code {rc}): {self.lib.ofx_last_error().decode()}")

    def grayscale_avg(self, src3):
        src3 = _c(src3, np.uint8)
        h, w, _ = src3.shape
        d = np.zeros_like(src3)
        self._f("grayscale_avg")(_p(src3, _u8p), _p(d, _u8p), h, w)  # (rows, cols) order, OptFlowGpu.cuh:5
        self._done("grayscale_avg")
        return d

    def conv_3ch(self, src3, mask, mw, mh, variant="conv_3ch_2d"):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.zeros_like(src3)
        self._f(variant)(_p(src3, _u8p), _p(d, _u8p), w, h, _p(mask, _f32p), mw, mh)
        self._done(variant)
        return d

    def conv_3ch_1ch(self, src3, mask, mw=3, mh=3, variant="conv_3ch_1ch_constant"):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.zeros((h, w), np.uint8)
        self._f(variant)(_p(src3, _u8p), w, h, _p(d, _u8p), _p(mask, _f32p), mw, mh)
        self._done(variant)
        return d

    def conv_3ch_1ch_float(self, src3, mask, mw=3, mh=3):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.zeros((h, w), np.float32)
        self._f("conv_3ch_1ch_tiled_uchar_float")(_p(src3, _u8p), w, h, _p(d, _f32p), _p(mask, _f32p), mw, mh)
        self._done("conv_3ch_1ch_tiled_uchar_float")
        return d

    def gauss_pyramid(self, img3, levels):
        img3 = _c(img3, np.uint8)
        h, w, _ = img3.shape
        pyr = [img3.copy()] + [np.zeros((h >> k, w >> k, 3), np.uint8) for k in range(1, levels)]
        self._f("gauss_pyramid")(_ptrs(pyr, C.c_uint8), w, h, levels, _p(self.GAUS_KERNEL_3x3, _f32p), 3, 3)
        self._done("gauss_pyramid")
        return pyr

    def srm_1ch(self, a, b, ww, wh, variant="srm_1ch"):
        a, b = _c(a, np.uint8), _c(b, np.uint8)
        h, w = a.shape
        d = np.zeros((h, w), np.int32)
        self._f(variant)(_p(a, _u8p), _p(b, _u8p), w, h, ww, wh, _p(d, _i32p))
        self._done(variant)
        return d

    def srm_1ch_float(self, a, b, ww, wh):
        a, b = _c(a, np.float32), _c(b, np.float32)
        h, w = a.shape
        d = np.zeros((h, w), np.float32)
        self._f("srm_1ch_float")(_p(a, _f32p), _p(b, _f32p), w, h, ww, wh, _p(d, _f32p))
        self._done("srm_1ch_float")
        return d

    def inverse_matrix(self, sxx, syy, sxy, sxt, syt):
        s = [_c(x, np.int32).copy() for x in (sxx, syy, sxy, sxt, syt)]
        h, w = s[0].shape
        flow = np.zeros((h, w, 2), np.float32)
        self._f("inverse_matrix")(*[_p(x, _i32p) for x in s], _ptrs([flow], C.c_float), 0, w, h)
        self._done("inverse_matrix")
        return flow

    def inverse_matrix_float(self, sxx, syy, sxy, sxt, syt):
        s = [_c(x, np.float32).copy() for x in (sxx, syy, sxy, sxt, syt)]
        h, w = s[0].shape
        flow = np.zeros((h, w, 2), np.float32)
        self._f("inverse_matrix_float")(*[_p(x, _f32p) for x in s], _ptrs([flow], C.c_float), 0, w, h)
        self._done("inverse_matrix_float")
        return flow

    def calc_opt_flow(self, prev3, next3, flow_pyr, level, max_level):
        prev3, next3 = _c(prev3, np.uint8), _c(next3, np.uint8).copy()
        h, w, _ = prev3.shape
        self._f("calc_opt_flow")(_p(prev3, _u8p), _p(next3, _u8p), w, h, _ptrs(flow_pyr, C.c_float), level, max_level)
        self._done("calc_opt_flow")
        return flow_pyr[level]

    def bilinear_filter(self, src3, gray3, ww, wh, sigma_s, sigma_b):
        src3, gray3 = _c(src3, np.uint8).copy(), _c(gray3, np.uint8).copy()
        h, w, _ = src3.shape
        d = np.zeros_like(src3)
        self._f("bilinear_filter")(_p(src3, _u8p), _p(gray3, _u8p), _p(d, _u8p), w, h, ww, wh, C.c_double(sigma_s), C.c_double(sigma_b))
        self._done("bilinear_filter")
        return d

    def flow_pair(self, prev3, next3, levels):
        """main.cu:246-262 with host buffers: pyramids of both frames, then calc_opt_flow coarse to fine."""
        pp, npyr = self.gauss_pyramid(prev3, levels), self.gauss_pyramid(next3, levels)
        h, w, _ = prev3.shape
        flow = [np.zeros((h >> k, w >> k, 2), np.float32) for k in range(levels)]
        for k in range(levels - 1, -1, -1):
            self.calc_opt_flow(pp[k], npyr[k], flow, k, levels)
        return flow, pp, npyr

    def frame_loop(self, w, h, levels):
        """main.cu's frame loop with ITS buffer discipline: the two image pyramids and the flow pyramid are allocated once
        (main.cu:203-205, alloc_pyramid) and reused for every frame; a frame is copied into level 0 of its pyramid (:246), the
        pyramid is built in place (:250), calc_opt_flow runs coarse to fine (:256-262), and the pyramids are swapped (:270-272).
        (flow_pair above allocates fresh numpy arrays per call: timing it measures first-touch page faults, not the library.)"""
        return _FrameLoop(self, w, h, levels)


    def conv_1d_3ch(self, src3):
        src3 = _c(src3, np.uint8).copy()
        h, w, _ = src3.shape
        d = np.zeros_like(src3)
        self._f("conv_1d_3ch")(_p(src3, _u8p), w, h, _p(d, _u8p))
        self._done("conv_1d_3ch")
        return d


class _FrameLoop:
    def __init__(self, gc, w, h, levels):
        self.gc, self.w, self.h, self.levels = gc, w, h, levels
        mk = lambda: [np.zeros((h >> k, w >> k, 3), np.uint8) for k in range(levels)]
        self.prev, self.cur = mk(), mk()
        self.flow = [np.zeros((h >> k, w >> k, 2), np.float32) for k in range(levels)]
        for a in self.prev + self.cur + self.flow:
            a.fill(0)   # touch every page once, as a running frame loop has
        self._gp, self._cof = gc._f("gauss_pyramid"), gc._f("calc_opt_flow")
        self._mask = _p(gc.GAUS_KERNEL_3x3, _f32p)
        self.have_prev = False

    def _pyramid(self, pyr, frame3):
        np.copyto(pyr[0], frame3)                                                     # main.cu:246
        self._gp(_ptrs(pyr, C.c_uint8), self.w, self.h, self.levels, self._mask, 3, 3)  # main.cu:250
        self.gc._done("gauss_pyramid")

    def first(self, frame3):
        self._pyramid(self.prev, frame3)                                              # main.cu:203-209
        self.have_prev = True

    def step(self, frame3):
        """one frame: its pyramid, every flow level against the previous frame's pyramid, swap; returns the flow pyramid (reused)"""
        assert self.have_prev
        self._pyramid(self.cur, frame3)
        fl = _ptrs(self.flow, C.c_float)
        for k in range(self.levels - 1, -1, -1):                                      # main.cu:256-262
            self._cof(_p(self.prev[k], _u8p), _p(self.cur[k], _u8p), self.w >> k, self.h >> k, fl, k, self.levels)
            self.gc._done("calc_opt_flow")
        self.prev, self.cur = self.cur, self.prev                                     # main.cu:270-272
        return self.flow


class UtilsCompat:
    """namespace utils of libofx_hip.so (include/OptFlowUtils.hpp) by mangled name."""

    def __init__(self):
        self.lib = _lib.load()

    def _f(self, name):
        f = getattr(self.lib, UTILS_SYMBOLS[name])
        f.restype = None
        return f

    def cleanup_outliers(self, img1):
        d = _c(img1, np.uint8).copy()
        h, w = d.shape
        self._f("cleanup_outliers")(_p(d, _u8p), w, h)
        return d

    def upscale(self, src, n):
        src = _c(src, np.uint8).copy()
        h, w = src.shape[:2]
        d = np.zeros((h << n, w << n) + ((3,) if src.ndim == 3 else ()), np.uint8)
        self._f("upscale_3ch" if src.ndim == 3 else "upscale_1ch")(_p(src, _u8p), w, h, n, _p(d, _u8p))
        return d

    def generate_gaussian_kernel(self, sigma, ks):
        n = ks if ks % 2 else ks + 1
        d = np.zeros((n, n), np.float64)
        self._f("generate_gaussian_kernel")(C.c_double(sigma), C.c_int(ks), d.ctypes.data_as(C.POINTER(C.c_double)))
        return d


class CpuCompat:
    """namespace cpu of libofx_hip.so (include/OptFlowCpu.hpp): the reference's CPU call surface, executed on the device."""

    def __init__(self):
        self.lib = _lib.load()

    def _f(self, name):
        f = getattr(self.lib, CPU_SYMBOLS[name])
        f.restype = None
        return f

    def _done(self, what):
        rc = self.lib.gpu_compat_last_status()
        if rc != 0:
            raise _lib.OfxError(f"cpu::{what} failed (code {rc}): {self.lib.ofx_last_error().decode()}")

    def sub_arr(self, a, b):
        a, b = _c(a, np.uint8).copy(), _c(b, np.uint8).copy()
        d = np.zeros_like(a)
        self._f("sub_arr")(_p(a, _u8p), _p(b, _u8p), a.size, _p(d, _u8p))
        self._done("sub_arr")
        return d

    def grayscale_avg_cpu(self, src3):
        src3 = _c(src3, np.uint8)
        h, w, _ = src3.shape
        d = np.zeros_like(src3)
        self._f("grayscale_avg_cpu")(_p(src3, _u8p), _p(d, _u8p), w, h)
        self._done("grayscale_avg_cpu")
        return d

    def conv_3ch(self, src3, mask, mw, mh):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.zeros_like(src3)
        self._f("conv_3ch")(_p(src3, _u8p), _p(mask, _f32p), _p(d, _u8p), w, h, mw, mh)
        self._done("conv_3ch")
        return d

    def conv_3ch_to_1ch(self, src3, mask, mw=3, mh=3):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.zeros((h, w), np.uint8)
        self._f("conv_3ch_to_1ch")(_p(src3, _u8p), w, h, _p(d, _u8p), _p(mask, _f32p), mw, mh)
        self._done("conv_3ch_to_1ch")
        return d

    def downscale_gaussian(self, src3, mask, mw=3, mh=3):
        src3, mask = _c(src3, np.uint8).copy(), _c(mask, np.float32)
        sh, sw, _ = src3.shape
        d = np.zeros((sh >> 1, sw >> 1, 3), np.uint8)
        self._f("downscale_gaussian")(_p(src3, _u8p), sw >> 1, sh >> 1, _p(d, _u8p), _p(mask, _f32p), mw, mh)
        self._done("downscale_gaussian")
        return d

    def gauss_pyramid(self, img3, levels, mask, mw=3, mh=3):
        img3, mask = _c(img3, np.uint8), _c(mask, np.float32)
        h, w, _ = img3.shape
        pyr = [img3.copy()] + [np.zeros((h >> k, w >> k, 3), np.uint8) for k in range(1, levels)]
        self._f("gauss_pyramid")(_ptrs(pyr, C.c_uint8), w, h, levels, _p(mask, _f32p), mw, mh)
        self._done("gauss_pyramid")
        return pyr

    def srm_1ch(self, a, b, ww, wh):
        a, b = _c(a, np.uint8), _c(b, np.uint8)
        h, w = a.shape
        d = np.zeros((h, w), np.int32)
        self._f("srm_1ch")(_p(a, _u8p), _p(b, _u8p), w, h, ww, wh, _p(d, _i32p))
        self._done("srm_1ch")
        return d

    def srm_3ch(self, a3, b3, ww, wh):
        a3, b3 = _c(a3, np.uint8).copy(), _c(b3, np.uint8).copy()
        h, w, _ = a3.shape
        d = np.zeros((h, w, 3), np.int32)
        self._f("srm_3ch")(_p(a3, _u8p), _p(b3, _u8p), w, h, ww, wh, _p(d, _i32p))
        self._done("srm_3ch")
        return d

    def shift_back_pyramid(self, src3, level, max_level, flow_pyr, dest_init=None):
        src3 = _c(src3, np.uint8)
        h, w, _ = src3.shape
        d = np.zeros_like(src3) if dest_init is None else _c(dest_init, np.uint8).copy()
        fl = [_c(f, np.float32) if f is not None else np.zeros(2, np.float32) for f in flow_pyr]
        self._f("shift_back_pyramid")(_p(src3, _u8p), w, h, level, max_level, _ptrs(fl, C.c_float), _p(d, _u8p))
        self._done("shift_back_pyramid")
        return d

    def inverse_matrix(self, sxx, syy, sxy, sxt, syt):
        s = [_c(x, np.int32).copy() for x in (sxx, syy, sxy, sxt, syt)]
        h, w = s[0].shape
        flow = np.zeros((h, w, 2), np.float32)
        self._f("inverse_matrix")(*[_p(x, _i32p) for x in s], _ptrs([flow], C.c_float), 0, w, h)
        self._done("inverse_matrix")
        return flow

    def calc_optical_flow(self, prev3, next3, flow_pyr, level, max_level):
        prev3, next3 = _c(prev3, np.uint8), _c(next3, np.uint8).copy()
        h, w, _ = prev3.shape
        self._f("calc_optical_flow")(_p(prev3, _u8p), _p(next3, _u8p), w, h, _ptrs(flow_pyr, C.c_float), level, max_level)
        self._done("calc_optical_flow")
        return flow_pyr[level]

    def bilinear_filter_3ch(self, src3, gray3, ww, wh, sigma_s, sigma_b):
        src3, gray3 = _c(src3, np.uint8).copy(), _c(gray3, np.uint8).copy()
        h, w, _ = src3.shape
        d = np.zeros_like(src3)
        self._f("bilinear_filter_3ch")(_p(src3, _u8p), _p(gray3, _u8p), _p(d, _u8p), w, h, ww, wh, C.c_double(sigma_s), C.c_double(sigma_b))
        self._done("bilinear_filter_3ch")
        return d

    def flow_pair(self, prev3, next3, levels, mask):
        """main.cu:246-262 with the cpu:: alternates (main.cu:251,261) swapped in"""
        pp, npyr = self.gauss_pyramid(prev3, levels, mask), self.gauss_pyramid(next3, levels, mask)
        h, w, _ = prev3.shape
        flow = [np.zeros((h >> k, w >> k, 2), np.float32) for k in range(levels)]
        for k in range(levels - 1, -1, -1):
            self.calc_optical_flow(pp[k], npyr[k], flow, k, levels)
        return flow, pp, npyr
