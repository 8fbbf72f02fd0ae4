"""ctypes binding of oracle/libmetdet_oracle.so (the C port of the CPU oracle).

TEST INFRASTRUCTURE ONLY -- used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MDH_ORACLE_LIB: another build of the same file (the ASan / UBSan one of `make -C oracle san`, tests/test_sanitizers.py)
_SO = os.environ.get("MDH_ORACLE_LIB") or os.path.join(_HERE, "libmetdet_oracle.so")
_lib = None

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "metdet_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        for name in ("ocv_median3", "ocv_dilate3", "ocv_erode3"):
            getattr(L, name).argtypes = [u8p, u8p, C.c_int, C.c_int]
            getattr(L, name).restype = None
        L.ocv_hough_lines_p.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, i32p, C.c_int]
        L.ocv_hough_lines_p.restype = C.c_int
        L.o_lineset_nms.argtypes = [i32p, C.c_int, i32p, f64p]
        L.o_lineset_nms.restype = C.c_int
        L.om3_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double,
                                 C.c_double, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double]
        L.om3_create.restype = C.c_void_p
        L.om3_destroy.argtypes = [C.c_void_p]
        L.om3_update.argtypes = [C.c_void_p, u8p]
        L.om3_detect.argtypes = [C.c_void_p, i32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double)]
        L.om3_detect.restype = C.c_int
        L.om3_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double),
                              C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.om3_num_threads.restype = C.c_int
        L.om3_set_threads.argtypes = [C.c_int]
        L.om3_set_threads.restype = None
        # small frames and hundreds of host cores do not mix: cap the OpenMP team (bench.py raises it explicitly)
        L.om3_set_threads(max(1, min(8, len(os.sched_getaffinity(0)))))
        _lib = L
    return _lib


def median3(img):
    out = np.empty_like(img)
    lib().ocv_median3(np.ascontiguousarray(img), out, *img.shape)
    return out


def dilate3(img):
    out = np.empty_like(img)
    lib().ocv_dilate3(np.ascontiguousarray(img), out, *img.shape)
    return out


def erode3(img):
    out = np.empty_like(img)
    lib().ocv_erode3(np.ascontiguousarray(img), out, *img.shape)
    return out


def hough_lines_p(image, rho, theta, threshold, min_line_length, max_line_gap, cap: int = 1 << 16):
    """Same contract as m3_oracle.hough_lines_p (rho=1, theta=pi/180 only)."""
    assert rho == 1 and abs(theta - np.pi / 180) < 1e-12
    img = np.ascontiguousarray(image)
    buf = np.zeros((cap, 4), np.int32)
    n = lib().ocv_hough_lines_p(img, img.shape[0], img.shape[1], int(threshold), float(min_line_length),
                                float(max_line_gap), buf, cap)
    assert n <= cap
    return buf[:n].copy()


def lineset_nms(lines):
    lines = np.ascontiguousarray(lines, dtype=np.int32).reshape(-1, 4)
    out = np.zeros_like(lines)
    prob = np.zeros(len(lines), np.float64)
    k = lib().o_lineset_nms(lines, len(lines), out, prob)
    return out[:k].copy(), prob[:k].copy()


class CM3:
    """The C port of M3Oracle: update() / detect() up to raw Hough segments."""
    SENS = {"low": (2.0, 4.4, 7), "normal": (1.2, 3.6, 5), "high": (0.9, 3.0, 3)}

    def __init__(self, n, mask, roi, *, adaptive=True, init_value=5, sensitivity="normal", interval=2,
                 hough_threshold=10, hough_min_len=10, hough_max_gap=10, dy_mask=True, cap=4096):
        self.H, self.W = mask.shape
        a, b, init = self.SENS[sensitivity]
        init_thr = init if adaptive else init_value
        roi_arr = (C.c_int * 4)(*[int(v) for v in roi])
        self._L = lib()
        self._h = self._L.om3_create(self.H, self.W, int(n), roi_arr, int(interval), int(bool(adaptive)), int(init_thr),
                                     a, b, int(hough_threshold), float(hough_min_len), float(hough_max_gap),
                                     int(bool(dy_mask)), float(np.sum(mask)))
        self.cap = cap
        self._lines = np.zeros((cap, 4), np.int32)

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.om3_destroy(self._h)
            self._h = None

    def update(self, frame):
        self._L.om3_update(self._h, np.ascontiguousarray(frame))

    def detect(self):
        lit, dsum, gap = C.c_int(), C.c_double(), C.c_double()
        n = self._L.om3_detect(self._h, self._lines, self.cap, C.byref(lit), C.byref(dsum), C.byref(gap))
        self.lit_count, self.dst_sum, self.gap, self.lines_num = lit.value, dsum.value, gap.value, n
        return self._lines[:min(n, self.cap)].copy()

    def state(self):
        thr, thr_f, snr, sig = C.c_int(), C.c_double(), C.c_double(), C.c_double()
        self._L.om3_get(self._h, None, None, C.byref(thr), C.byref(thr_f), C.byref(snr), C.byref(sig))
        return thr.value, thr_f.value, snr.value, sig.value

    def dst(self):
        out = np.empty((self.H, self.W), np.uint8)
        self._L.om3_get(self._h, out.ctypes.data, None, None, None, None, None)
        return out

    def act(self):
        out = np.empty((self.H, self.W), np.uint8)
        self._L.om3_get(self._h, None, out.ctypes.data, None, None, None, None)
        return out
