"""ctypes bindings used by the tests: the CPU oracle (oracle/libamt_oracle.so) and, where built,
the real reference compiled through the shim (oracle/_ref/libamt_ref.so).

Checker side only -- the product library has its own binding in amatsukaze_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

c_f = C.c_float
c_i = C.c_int
c_i64 = C.c_int64
c_p = C.c_void_p
c_s = C.c_char_p
PF = C.POINTER(C.c_float)
PI = C.POINTER(C.c_int)


def _ptr(a):
    return a.ctypes.data_as(c_p) if a is not None else None


def _decl(lib, name, res, args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = args
    return f


def build_oracle():
    so = os.path.join(ROOT, "oracle", "libamt_oracle.so")
    src = os.path.join(ROOT, "oracle", "amt_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    return so


def ref_path():
    return os.path.join(ROOT, "oracle", "_ref", "libamt_ref.so")


class Oracle:
    def __init__(self):
        L = self.lib = C.CDLL(build_oracle())
        d = lambda n, r, a: _decl(L, n, r, a)
        d("orc_corr5x5_scalar", c_f, [c_p, c_p, c_i, c_i, c_i, c_p])
        d("orc_corr5x5_avx", c_f, [c_p, c_p, c_i, c_i, c_i, c_p])
        d("orc_logo_create", c_p, [c_i] * 8 + [c_p])
        d("orc_logo_load", c_p, [c_s])
        d("orc_logo_save", c_i, [c_p, c_s, c_s, c_i])
        d("orc_logo_free", None, [c_p])
        d("orc_logo_deint", c_p, [c_p])
        d("orc_logo_field", c_p, [c_p, c_i])
        d("orc_logo_info", None, [c_p, c_p])
        d("orc_logo_data", c_p, [c_p])
        d("orc_logo_create_mask", None, [c_p, c_f, c_i])
        d("orc_logo_mask", c_p, [c_p])
        d("orc_logo_kernels", c_p, [c_p])
        d("orc_logo_scales", c_p, [c_p])
        d("orc_logo_black_score", c_f, [c_p])
        d("orc_evaluate_logo", c_f, [c_p, c_p, c_f, c_f, c_p, c_i])
        d("orc_deint_y_u8", None, [c_p, c_p, c_i, c_i, c_i])
        d("orc_deint_y_u16", None, [c_p, c_p, c_i, c_i, c_i])
        d("orc_copy_y_u8", None, [c_p, c_p, c_i, c_i, c_i])
        d("orc_logoframe_scan", None, [c_p, c_i, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_p])
        d("orc_logoframe_select", None, [c_p, c_i, c_i, c_i, c_p, c_p])
        d("orc_logoframe_write_result", c_i, [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i])
        d("orc_analyze_frames", None, [c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p])
        d("orc_delogo_u8", None, [c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_f])
        d("orc_delogo_u16", None, [c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_f])
        d("orc_calc_fade2", None, [c_p, c_i, c_i, c_p, c_p])
        d("orc_calc_fade", None, [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_p])
        d("orc_read_logoframe", c_i, [c_s, c_i, c_p])
        d("orc_erase_frame", None, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f])
        d("orc_scan_create", c_p, [c_i] * 5)
        d("orc_scan_free", None, [c_p])
        d("orc_scan_add_frame_u8", c_i, [c_p, c_p, c_p, c_p, c_i, c_i])
        d("orc_scan_nframes", c_i, [c_p])
        d("orc_scan_sums", None, [c_p, c_p])
        d("orc_scan_get_logo", c_p, [c_p, c_i, c_i, c_i, c_i, c_i, c_i])
        d("orc_frame_metrics", None, [c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_p, c_p])
        d("orc_merge_field", None, [c_p] * 6 + [c_i] * 6 + [c_p] * 3 + [c_i] * 2)
        d("orc_scanlogo", c_p, [c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p])
        d("orc_scanlogo_mt", c_p, [c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_p])

    # ---- helpers ----
    def logo_info(self, h):
        o = np.zeros(10, np.int32)
        self.lib.orc_logo_info(h, _ptr(o))
        return o

    def logo_arrays(self, h):
        """(data, mask, kernels[count*25], scales[count*32*2], blackScore, maskpixels, count)"""
        o = self.logo_info(h)
        w, hh, lx, ly = int(o[0]), int(o[1]), int(o[2]), int(o[3])
        nd = (w * hh + 2 * (w >> lx) * (hh >> ly)) * 2
        data = np.ctypeslib.as_array(C.cast(self.lib.orc_logo_data(h), PF), (nd,)).copy()
        mp, cnt = int(o[8]), int(o[9])
        if mp == 0:
            return data, None, None, None, 0.0, 0, 0
        mask = np.ctypeslib.as_array(C.cast(self.lib.orc_logo_mask(h), C.POINTER(C.c_uint8)), (w * hh,)).copy()
        ker = np.ctypeslib.as_array(C.cast(self.lib.orc_logo_kernels(h), PF), (mp * 25,)).copy()
        sc = np.ctypeslib.as_array(C.cast(self.lib.orc_logo_scales(h), PF), (mp * 64,)).copy()
        return data, mask, ker, sc, float(self.lib.orc_logo_black_score(h)), mp, cnt

    def make_logo(self, data, w, h, imgw, imgh, imgx, imgy):
        data = np.ascontiguousarray(data, np.float32)
        return self.lib.orc_logo_create(w, h, 1, 1, imgw, imgh, imgx, imgy, _ptr(data))


class Ref:
    """The real reference (oracle/_ref/libamt_ref.so); available() is False when it was never built."""

    @staticmethod
    def available():
        return os.path.exists(ref_path())

    def __init__(self):
        L = self.lib = C.CDLL(ref_path())
        d = lambda n, r, a: _decl(L, n, r, a)
        d("ref_last_error", c_s, [])
        d("ref_is_avx", c_i, [])
        d("ref_corr5x5_scalar", c_f, [c_p, c_p, c_i, c_i, c_i, c_p])
        d("ref_corr5x5_avx", c_f, [c_p, c_p, c_i, c_i, c_i, c_p])
        d("ref_logo_load", c_p, [c_s])
        d("ref_logo_save", c_i, [c_p, c_s])
        d("ref_logo_free", None, [c_p])
        d("ref_logo_deint", c_p, [c_p])
        d("ref_logo_field", c_p, [c_p, c_i])
        d("ref_logo_create_mask", None, [c_p, c_f])
        d("ref_logo_info", None, [c_p, c_p])
        d("ref_logo_data", c_p, [c_p])
        d("ref_logo_mask", c_p, [c_p])
        d("ref_logo_kernels", c_p, [c_p])
        d("ref_logo_scales", c_p, [c_p])
        d("ref_logo_black_score", c_f, [c_p])
        d("ref_evaluate_logo", c_f, [c_p, c_p, c_f, c_f, c_p, c_i])
        d("ref_deint_y_u8", None, [c_p, c_p, c_i, c_i, c_i])
        d("ref_deint_y_u16", None, [c_p, c_p, c_i, c_i, c_i])
        d("ref_logoframe", c_i, [c_p, c_i, c_f, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_i, c_s, c_p, c_i])
        d("ref_analyze", c_i, [c_s, c_f, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_p])
        d("ref_erase", c_i, [c_s, c_s, c_i, c_f, c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_p])
        d("ref_scan_create", c_p, [c_i] * 5)
        d("ref_scan_free", None, [c_p])
        d("ref_scan_add_frame_u8", c_i, [c_p, c_p, c_p, c_p, c_i, c_i])
        d("ref_scan_nframes", c_i, [c_p])
        d("ref_scan_sums", None, [c_p, c_p])
        d("ref_scan_get_logo", c_p, [c_p, c_i, c_i, c_i, c_i, c_i, c_i])
        d("ref_scanlogo", c_i, [c_s, c_i, c_s, c_s, c_i, c_i, c_i, c_i, c_i, c_i])

    def logo_info(self, h):
        o = np.zeros(10, np.int32)
        self.lib.ref_logo_info(h, _ptr(o))
        return o

    def logo_data(self, h):
        o = self.logo_info(h)
        w, hh, lx, ly = int(o[0]), int(o[1]), int(o[2]), int(o[3])
        nd = (w * hh + 2 * (w >> lx) * (hh >> ly)) * 2
        return np.ctypeslib.as_array(C.cast(self.lib.ref_logo_data(h), PF), (nd,)).copy()

    def logo_tables(self, h, count):
        o = self.logo_info(h)
        w, hh, mp = int(o[0]), int(o[1]), int(o[8])
        mask = np.ctypeslib.as_array(C.cast(self.lib.ref_logo_mask(h), C.POINTER(C.c_uint8)), (w * hh,)).copy()
        ker = np.ctypeslib.as_array(C.cast(self.lib.ref_logo_kernels(h), PF), (mp * 25,)).copy()
        sc = np.ctypeslib.as_array(C.cast(self.lib.ref_logo_scales(h), PF), (mp * 64,)).copy()
        return mask, ker[:count * 25], sc[:count * 64], float(self.lib.ref_logo_black_score(h)), mp


def write_raw_clip(path, Y, U, V, W, H):
    """raw clip file the shim 'decoder' reads: int32 {'AMTR', w, h, n} + tight Y,U,V frames (8-bit)."""
    n = Y.shape[0]
    with open(path, "wb") as f:
        f.write(np.array([0x52544D41, W, H, n], np.int32).tobytes())
        for i in range(n):
            f.write(np.ascontiguousarray(Y[i, :, :W]).tobytes())
            f.write(np.ascontiguousarray(U[i, :, :W // 2]).tobytes())
            f.write(np.ascontiguousarray(V[i, :, :W // 2]).tobytes())
