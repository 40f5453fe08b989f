"""CPU ORACLE loader -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (``cuda_optical_flow_2_amd``)
never does; it fails loudly when its HIP library is missing instead of falling
back to anything here.

Two libraries are exposed through ctypes + numpy:

* ``Oracle``  -> ``oracle/liboracle.so``: this repo's plain-C restatement
  (``ofx_oracle.c``), buildable anywhere with gcc.
* ``Reference`` -> ``oracle/_ref/libref_cpu.so``: the reference's OWN CPU
  sources (OptFlowCPU.cpp, kernels.cpp, OptFlowUtils.cpp) compiled in the build
  container by ``oracle/Makefile`` (target ``ref``).  Present only when it was
  built there; it travels to the GPU box as a prebuilt file.  Functions are
  bound by their Itanium-mangled C++ names.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OFX_ORACLE_SO: another build of the same restatement (liboracle_ubsan.so, `make -C oracle ubsan`)
ORACLE_SO = os.path.join(_HERE, os.environ.get("OFX_ORACLE_SO", "liboracle.so"))
REF_SO = os.path.join(_HERE, "_ref", "libref_cpu.so")

_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)


def build(force: bool = False) -> None:
    """Compile liboracle.so, and _ref/libref_cpu.so when /root/reference exists."""
    src = os.path.join(_HERE, "ofx_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference") and (force or not os.path.exists(REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "ref"], stdout=subprocess.DEVNULL)


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _c(a, dtype):
    a = np.ascontiguousarray(a, dtype=dtype)
    return a


def _ptr_array(levels, ctype):
    arr = (C.POINTER(ctype) * len(levels))()
    for i, lv in enumerate(levels):
        arr[i] = lv.ctypes.data_as(C.POINTER(ctype))
    return arr


class Oracle:
    """numpy front-end to liboracle.so (ofx_oracle.h)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build()
        self.lib = C.CDLL(ORACLE_SO)
        L = self.lib
        for name in ("orc_Dx_3x3", "orc_Dy_3x3", "orc_Dt_3x3", "orc_GAUS_3x3"):
            setattr(self, name[4:], np.array((C.c_float * 9).in_dll(L, name), dtype=np.float32))

    # ---- element-wise
    def grayscale_avg(self, src3):
        src3 = _c(src3, np.uint8)
        h, w, _ = src3.shape
        dst = np.empty_like(src3)
        self.lib.orc_grayscale_avg(_p(src3, _u8p), _p(dst, _u8p), w, h)
        return dst

    def sub_u8(self, a, b):
        a, b = _c(a, np.uint8), _c(b, np.uint8)
        d = np.empty_like(a)
        self.lib.orc_sub_u8(_p(a, _u8p), _p(b, _u8p), a.size, _p(d, _u8p))
        return d

    # ---- correlations
    def conv_3ch(self, src3, mask, mw, mh):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.empty_like(src3)
        self.lib.orc_conv_3ch(_p(src3, _u8p), _p(mask, _f32p), _p(d, _u8p), w, h, mw, mh)
        return d

    def conv_3ch_to_1ch(self, src3, mask, mw=3, mh=3):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.empty((h, w), np.uint8)
        self.lib.orc_conv_3ch_to_1ch(_p(src3, _u8p), w, h, _p(d, _u8p), _p(mask, _f32p), mw, mh)
        return d

    def conv_3ch_to_1ch_f32(self, src3, mask, mw=3, mh=3):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.empty((h, w), np.float32)
        self.lib.orc_conv_3ch_to_1ch_f32(_p(src3, _u8p), w, h, _p(d, _f32p), _p(mask, _f32p), mw, mh)
        return d

    # ---- pyramid
    def downscale_gaussian(self, src3, mask=None, mw=3, mh=3):
        src3 = _c(src3, np.uint8)
        mask = self.GAUS_3x3 if mask is None else _c(mask, np.float32)
        sh, sw, _ = src3.shape
        h, w = sh >> 1, sw >> 1
        d = np.empty((h, w, 3), np.uint8)
        self.lib.orc_downscale_gaussian(_p(src3, _u8p), w, h, _p(d, _u8p), _p(mask, _f32p), mw, mh)
        return d

    def gauss_pyramid(self, img3, levels, mask=None, mw=3, mh=3):
        img3 = _c(img3, np.uint8)
        mask = self.GAUS_3x3 if mask is None else _c(mask, np.float32)
        h, w, _ = img3.shape
        pyr = [img3.copy()] + [np.empty((h >> k, w >> k, 3), np.uint8) for k in range(1, levels)]
        arr = _ptr_array(pyr, C.c_uint8)
        self.lib.orc_gauss_pyramid(arr, w, h, levels, _p(mask, _f32p), mw, mh)
        return pyr

    # ---- window sums
    def srm_1ch(self, a, b, ww, wh):
        a, b = _c(a, np.uint8), _c(b, np.uint8)
        h, w = a.shape
        d = np.empty((h, w), np.int32)
        self.lib.orc_srm_1ch(_p(a, _u8p), _p(b, _u8p), w, h, ww, wh, _p(d, _i32p))
        return d

    def srm_1ch_f32(self, a, b, ww, wh, exact=False):
        a, b = _c(a, np.float32), _c(b, np.float32)
        h, w = a.shape
        d = np.empty((h, w), np.float32)
        fn = self.lib.orc_srm_1ch_f32_exact if exact else self.lib.orc_srm_1ch_f32
        fn(_p(a, _f32p), _p(b, _f32p), w, h, ww, wh, _p(d, _f32p))
        return d

    def srm_3ch(self, a3, b3, ww, wh):
        a3, b3 = _c(a3, np.uint8), _c(b3, np.uint8)
        h, w, _ = a3.shape
        d = np.empty((h, w, 3), np.int32)
        self.lib.orc_srm_3ch(_p(a3, _u8p), _p(b3, _u8p), w, h, ww, wh, _p(d, _i32p))
        return d

    # ---- link-compat leftovers (utils::, gpu::conv_1d_3ch)
    def cleanup_outliers(self, img1):
        d = _c(img1, np.uint8).copy()
        h, w = d.shape
        self.lib.orc_cleanup_outliers(_p(d, _u8p), w, h)
        return d

    def upscale(self, src, n):
        src = _c(src, np.uint8)
        ch = 1 if src.ndim == 2 else src.shape[2]
        h, w = src.shape[:2]
        d = np.empty((h << n, w << n) + ((ch,) if src.ndim == 3 else ()), np.uint8)
        self.lib.orc_upscale(_p(src, _u8p), w, h, n, ch, _p(d, _u8p))
        return d

    def conv_1d_3ch(self, src3):
        src3 = _c(src3, np.uint8)
        h, w, _ = src3.shape
        d = np.empty_like(src3)
        self.lib.orc_conv_1d_3ch(_p(src3, _u8p), w, h, _p(d, _u8p))
        return d

    # ---- shift
    def shift_back_pyramid(self, src3, level, max_level, flow_pyr, dest_init=None):
        """dest_init: what the caller's destination buffer holds before the call (default zeros)"""
        src3 = _c(src3, np.uint8)
        h, w, _ = src3.shape
        d = np.zeros_like(src3) if dest_init is None else _c(dest_init, np.uint8).copy()
        fl = [_c(f, np.float32) if f is not None else np.zeros(2, np.float32) for f in flow_pyr]
        self.lib.orc_shift_back_pyramid(_p(src3, _u8p), w, h, level, max_level, _ptr_array(fl, C.c_float), _p(d, _u8p))
        return d

    # ---- solves
    def _solve(self, fn, sums, isint):
        dt, pt = (np.int32, _i32p) if isint else (np.float32, _f32p)
        s = [_c(x, dt) for x in sums]
        h, w = s[0].shape
        flow = np.empty((h, w, 2), np.float32)
        fn(*[_p(x, pt) for x in s], _p(flow, _f32p), w, h)
        return flow

    def inverse_matrix_f32arith(self, sxx, syy, sxy, sxt, syt):
        return self._solve(self.lib.orc_inverse_matrix_f32arith, (sxx, syy, sxy, sxt, syt), True)

    def inverse_matrix_i32(self, sxx, syy, sxy, sxt, syt):
        return self._solve(self.lib.orc_inverse_matrix_i32, (sxx, syy, sxy, sxt, syt), True)

    def inverse_matrix_f32(self, sxx, syy, sxy, sxt, syt):
        return self._solve(self.lib.orc_inverse_matrix_f32, (sxx, syy, sxy, sxt, syt), False)

    def inverse_matrix_inline_cpu(self, sxx, syy, sxy, sxt, syt):
        return self._solve(self.lib.orc_inverse_matrix_inline_cpu, (sxx, syy, sxy, sxt, syt), True)

    # ---- level compositions; flow_pyr is a list of (h_k, w_k, 2) float32 arrays, updated in place
    def calc_optical_flow_cpu(self, prev3, next3, flow_pyr, level, max_level, window=9):
        prev3, next3 = _c(prev3, np.uint8), _c(next3, np.uint8)
        h, w, _ = prev3.shape
        self.lib.orc_calc_optical_flow_cpu(_p(prev3, _u8p), _p(next3, _u8p), w, h,
                                           _ptr_array(flow_pyr, C.c_float), level, max_level, window)
        return flow_pyr[level]

    def calc_opt_flow_gpu(self, prev3, next3, flow_pyr, level, max_level, window=19, exact_sums=False):
        prev3, next3 = _c(prev3, np.uint8), _c(next3, np.uint8)
        h, w, _ = prev3.shape
        self.lib.orc_calc_opt_flow_gpu(_p(prev3, _u8p), _p(next3, _u8p), w, h,
                                       _ptr_array(flow_pyr, C.c_float), level, max_level, window, int(exact_sums))
        return flow_pyr[level]

    def level_planes(self, prev3, next3, window, mode, exact_sums=False, want_sums=True):
        prev3, next3 = _c(prev3, np.uint8), _c(next3, np.uint8)
        h, w, _ = prev3.shape
        ix, iy, it = (np.empty((h, w), np.float32) for _ in range(3))
        sums = np.empty((5, h, w), np.float64) if want_sums else None
        self.lib.orc_level_planes(_p(prev3, _u8p), _p(next3, _u8p), w, h, window, mode, int(exact_sums),
                                  _p(ix, _f32p), _p(iy, _f32p), _p(it, _f32p), _p(sums, _f64p))
        return ix, iy, it, sums

    def compose_flow(self, flow_pyr, levels, level):
        h, w, _ = flow_pyr[level].shape
        d = np.empty((h, w, 2), np.float32)
        self.lib.orc_compose_flow(_ptr_array(flow_pyr, C.c_float), w, h, levels, level, _p(d, _f32p))
        return d

    # ---- whole pair, the sequence main.cu:246-262 runs (pyramid of both frames, then coarse->fine)
    def flow_pair(self, prev3, next3, levels, window, mode, exact_sums=False):
        """mode 'compat_cpu' or 'lk_float'.  Returns (flow_pyr, prev_pyr, next_pyr)."""
        pp = self.gauss_pyramid(prev3, levels)
        npyr = self.gauss_pyramid(next3, levels)
        h, w, _ = prev3.shape
        flow = [np.zeros((h >> k, w >> k, 2), np.float32) for k in range(levels)]
        for k in range(levels - 1, -1, -1):
            if mode == "compat_cpu":
                self.calc_optical_flow_cpu(pp[k], npyr[k], flow, k, levels, window)
            else:
                self.calc_opt_flow_gpu(pp[k], npyr[k], flow, k, levels, window, exact_sums)
        return flow, pp, npyr

    # ---- extension: iterative refinement (no reference twin; DESIGN.md "lk_iter")
    ITER_SCALE = np.float32(8.0 / 15.0)

    def warp_bilinear_u8(self, src1, flow, scale=None):
        src1, flow = _c(src1, np.uint8), _c(flow, np.float32)
        h, w = src1.shape
        d = np.empty_like(src1)
        self.lib.orc_warp_bilinear_u8(_p(src1, _u8p), w, h, _p(flow, _f32p), C.c_float(self.ITER_SCALE if scale is None else scale), _p(d, _u8p))
        return d

    def lk_iter_level(self, prev1, next1_shifted, window, iters):
        prev1, next1_shifted = _c(prev1, np.uint8), _c(next1_shifted, np.uint8)
        h, w = prev1.shape
        flow = np.empty((h, w, 2), np.float32)
        first = np.empty((h, w, 2), np.float32)
        self.lib.orc_lk_iter_level(_p(prev1, _u8p), _p(next1_shifted, _u8p), w, h, window, iters, _p(flow, _f32p), _p(first, _f32p))
        return flow, first

    def flow_pair_iter(self, prev1, next1, levels, window, iters):
        """lk_float pyramid with `iters` refinement iterations per level.  Returns the flow pyramid (accumulated)."""
        pp = self.gauss_pyramid(np.repeat(prev1[:, :, None], 3, 2), levels)
        npyr = self.gauss_pyramid(np.repeat(next1[:, :, None], 3, 2), levels)
        h, w = prev1.shape
        first = [np.zeros((h >> k, w >> k, 2), np.float32) for k in range(levels)]
        out = [None] * levels
        for k in range(levels - 1, -1, -1):
            nxt3 = npyr[k] if k == levels - 1 else self.shift_back_pyramid(npyr[k], k, levels, first)
            out[k], first[k] = self.lk_iter_level(pp[k][:, :, 0], nxt3[:, :, 0], window, iters)
        return out

    # ---- bilateral
    def generate_gaussian_kernel(self, sigma, ks):
        n = ks if ks % 2 else ks + 1
        d = np.empty((n, n), np.float64)
        self.lib.orc_generate_gaussian_kernel(C.c_double(sigma), ks, _p(d, _f64p))
        return d

    def bilateral_3ch(self, src3, gray3, ww, wh, sigma_s, sigma_b):
        src3, gray3 = _c(src3, np.uint8), _c(gray3, np.uint8)
        h, w, _ = src3.shape
        d = np.empty_like(src3)
        self.lib.orc_bilateral_3ch(_p(src3, _u8p), _p(gray3, _u8p), _p(d, _u8p), w, h, ww, wh,
                                   C.c_double(sigma_s), C.c_double(sigma_b))
        return d


class Reference:
    """The reference's own CPU build (oracle/_ref/libref_cpu.so), by mangled name."""

    def __init__(self):
        if not os.path.exists(REF_SO):
            raise FileNotFoundError(REF_SO + " (built only where /root/reference exists: make -C oracle ref)")
        self.lib = C.CDLL(REF_SO)
        for name in ("Dx_3x3", "Dy_3x3", "Dt_3x3", "GAUS_KERNEL_3x3"):
            setattr(self, name, np.array((C.c_float * 9).in_dll(self.lib, name), dtype=np.float32))
        self.GAUS_KERNEL_5x5 = np.array((C.c_float * 25).in_dll(self.lib, "GAUS_KERNEL_5x5"), dtype=np.float32)

    def _f(self, mangled):
        f = getattr(self.lib, mangled)
        f.restype = None
        return f

    def grayscale_avg_cpu(self, src3):
        src3 = _c(src3, np.uint8)
        h, w, _ = src3.shape
        d = np.empty_like(src3)
        self._f("_ZN3cpu17grayscale_avg_cpuEPKhPhii")(_p(src3, _u8p), _p(d, _u8p), w, h)
        return d

    def sub_arr(self, a, b):
        a, b = _c(a, np.uint8).copy(), _c(b, np.uint8).copy()
        d = np.empty_like(a)
        self._f("_ZN3cpu7sub_arrEPhS0_iS0_")(_p(a, _u8p), _p(b, _u8p), a.size, _p(d, _u8p))
        return d

    def conv_3ch(self, src3, mask, mw, mh):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.empty_like(src3)
        self._f("_ZN3cpu8conv_3chEPKhPKfPhiiii")(_p(src3, _u8p), _p(mask, _f32p), _p(d, _u8p), w, h, mw, mh)
        return d

    def conv_3ch_to_1ch(self, src3, mask, mw=3, mh=3):
        src3, mask = _c(src3, np.uint8), _c(mask, np.float32)
        h, w, _ = src3.shape
        d = np.empty((h, w), np.uint8)
        self._f("_ZN3cpu15conv_3ch_to_1chEPKhiiPhPKfii")(_p(src3, _u8p), w, h, _p(d, _u8p), _p(mask, _f32p), mw, mh)
        return d

    def downscale_gaussian(self, src3, mask=None, mw=3, mh=3):
        src3 = _c(src3, np.uint8).copy()
        mask = self.GAUS_KERNEL_3x3 if mask is None else _c(mask, np.float32)
        sh, sw, _ = src3.shape
        h, w = sh >> 1, sw >> 1
        d = np.empty((h, w, 3), np.uint8)
        self._f("_ZN3cpu18downscale_gaussianEPhiiS0_PKfii")(_p(src3, _u8p), w, h, _p(d, _u8p), _p(mask, _f32p), mw, mh)
        return d

    def gauss_pyramid(self, img3, levels, mask=None, mw=3, mh=3):
        img3 = _c(img3, np.uint8)
        mask = self.GAUS_KERNEL_3x3 if mask is None else _c(mask, np.float32)
        h, w, _ = img3.shape
        pyr = [img3.copy()] + [np.empty((h >> k, w >> k, 3), np.uint8) for k in range(1, levels)]
        self._f("_ZN3cpu13gauss_pyramidEPPhiiiPKfii")(_ptr_array(pyr, C.c_uint8), w, h, levels, _p(mask, _f32p), mw, mh)
        return pyr

    def srm_3ch(self, a3, b3, ww, wh):
        """cpu::srm_3ch reads up to one row + one pixel past its inputs (bounds test `>`, OptFlowCPU.cpp:222): the inputs
        are handed over inside larger zero-filled buffers, so that those reads are defined (and contribute nothing)."""
        a3, b3 = _c(a3, np.uint8), _c(b3, np.uint8)
        h, w, _ = a3.shape
        pa, pb = np.zeros((h + 2, w, 3), np.uint8), np.zeros((h + 2, w, 3), np.uint8)
        pa[:h], pb[:h] = a3, b3
        d = np.empty((h, w, 3), np.int32)
        self._f("_ZN3cpu7srm_3chEPhS0_iiiiPi")(_p(pa, _u8p), _p(pb, _u8p), w, h, ww, wh, _p(d, _i32p))
        return d

    def cleanup_outliers(self, img1):
        d = _c(img1, np.uint8).copy()
        h, w = d.shape
        self._f("_ZN5utils16cleanup_outliersEPhii")(_p(d, _u8p), w, h)
        return d

    def upscale(self, src, n):
        src = _c(src, np.uint8).copy()
        h, w = src.shape[:2]
        if src.ndim == 3:
            d = np.empty((h << n, w << n, 3), np.uint8)
            self._f("_ZN5utils11upscale_3chEPhiiiS0_")(_p(src, _u8p), w, h, n, _p(d, _u8p))
        else:
            d = np.empty((h << n, w << n), np.uint8)
            self._f("_ZN5utils11upscale_1chEPhiiiS0_")(_p(src, _u8p), w, h, n, _p(d, _u8p))
        return d

    def srm_1ch(self, a, b, ww, wh):
        a, b = _c(a, np.uint8), _c(b, np.uint8)
        h, w = a.shape
        d = np.empty((h, w), np.int32)
        self._f("_ZN3cpu7srm_1chEPKhS1_iiiiPi")(_p(a, _u8p), _p(b, _u8p), w, h, ww, wh, _p(d, _i32p))
        return d

    def shift_back_pyramid(self, src3, level, max_level, flow_pyr, dest_init=None):
        src3 = _c(src3, np.uint8)
        h, w, _ = src3.shape
        d = np.zeros_like(src3) if dest_init is None else _c(dest_init, np.uint8).copy()
        fl = [_c(f, np.float32) if f is not None else np.zeros(2, np.float32) for f in flow_pyr]
        self._f("_ZN3cpu18shift_back_pyramidEPKhiiiiPPfPh")(_p(src3, _u8p), w, h, level, max_level,
                                                           _ptr_array(fl, C.c_float), _p(d, _u8p))
        return d

    def inverse_matrix(self, sxx, syy, sxy, sxt, syt):
        s = [_c(x, np.int32).copy() for x in (sxx, syy, sxy, sxt, syt)]
        h, w = s[0].shape
        flow = np.empty((h, w, 2), np.float32)
        self._f("_ZN3cpu14inverse_matrixEPiS0_S0_S0_S0_PPfiii")(*[_p(x, _i32p) for x in s],
                                                               _ptr_array([flow], C.c_float), 0, w, h)
        return flow

    def calc_optical_flow(self, prev3, next3, flow_pyr, level, max_level):
        prev3, next3 = _c(prev3, np.uint8), _c(next3, np.uint8).copy()
        h, w, _ = prev3.shape
        self._f("_ZN3cpu17calc_optical_flowEPKhPhiiPPfii")(_p(prev3, _u8p), _p(next3, _u8p), w, h,
                                                          _ptr_array(flow_pyr, C.c_float), level, max_level)
        return flow_pyr[level]

    def flow_pair(self, prev3, next3, levels):
        pp = self.gauss_pyramid(prev3, levels)
        npyr = self.gauss_pyramid(next3, levels)
        h, w, _ = prev3.shape
        flow = [np.zeros((h >> k, w >> k, 2), np.float32) for k in range(levels)]
        for k in range(levels - 1, -1, -1):
            self.calc_optical_flow(pp[k], npyr[k], flow, k, levels)
        return flow, pp, npyr

    def generate_gaussian_kernel(self, sigma, ks):
        n = ks if ks % 2 else ks + 1
        d = np.empty((n, n), np.float64)
        f = self._f("_ZN5utils24generate_gaussian_kernelEdiPd")
        f(C.c_double(sigma), C.c_int(ks), _p(d, _f64p))
        return d

    def bilinear_filter_3ch(self, src3, gray3, ww, wh, sigma_s, sigma_b):
        src3, gray3 = _c(src3, np.uint8).copy(), _c(gray3, np.uint8).copy()
        h, w, _ = src3.shape
        d = np.empty_like(src3)
        f = self._f("_ZN3cpu19bilinear_filter_3chEPhS0_S0_iiiidd")
        f(_p(src3, _u8p), _p(gray3, _u8p), _p(d, _u8p), w, h, ww, wh, C.c_double(sigma_s), C.c_double(sigma_b))
        return d


def have_reference() -> bool:
    return os.path.exists(REF_SO)
