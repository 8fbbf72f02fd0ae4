#!/usr/bin/env python3
"""Throughput of the loader-side kernels with inputs resident in HBM:
   preprocess: 3840x2160x3 BGR -> resize 960x540 (INTER_LINEAR) -> gray -> mask -> max over exp_frame frames
   max_stack : running max over a clip of K colour frames.
Prints one JSON line; the CPU figure beside it is the NumPy oracle restatement on one host core."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metdetpy_amd import _lib
from metdetpy_amd.imgproc import Preprocessor
from oracle import m3_oracle as mo


def hip():
    h = C.CDLL("libamdhip64.so")
    h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    h.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    h.hipFree.argtypes = [C.c_void_p]
    return h


def main():
    L = _lib.load()
    h = hip()
    sh, sw, dw, dh, exp = 2160, 3840, 960, 540, 3
    rng = np.random.default_rng(1)
    src = rng.integers(0, 256, (exp, sh, sw, 3), dtype=np.uint8)
    mask = np.ones((dh, dw), np.uint8)
    pre = Preprocessor((sh, sw), (dw, dh), mask)
    d_src, d_out = C.c_void_p(), C.c_void_p()
    h.hipMalloc(C.byref(d_src), src.nbytes)
    h.hipMalloc(C.byref(d_out), dh * dw)
    h.hipMemcpy(d_src, src.ctypes.data, src.nbytes, 1)
    for _ in range(3):
        pre.run_device(d_src.value, exp, d_out.value)
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        pre.run_device(d_src.value, exp, d_out.value)
    dt = (time.perf_counter() - t0) / reps
    got = np.empty((dh, dw), np.uint8)
    h.hipMemcpy(got.ctypes.data, d_out, dh * dw, 2)
    t1 = time.perf_counter()
    want = mo.merge_max([mo.preprocess_frame(f, (dw, dh), mask) for f in src])
    cpu_dt = time.perf_counter() - t1
    out = {"preprocess": {"src": [sw, sh, 3], "dst": [dw, dh], "exp_frame": exp, "ms_per_output_frame": dt * 1e3,
                          "output_frames_per_s": 1 / dt, "source_GBps": src.nbytes / dt / 1e9,
                          "parity_vs_oracle": bool(np.array_equal(got, want)),
                          "cpu_numpy_oracle_ms": cpu_dt * 1e3}}
    # clip max-stack: 64 colour frames at 1920x1080
    K, fb = 64, 1080 * 1920 * 3
    clip = rng.integers(0, 256, (K, 1080, 1920, 3), dtype=np.uint8)
    d_clip, d_acc = C.c_void_p(), C.c_void_p()
    h.hipMalloc(C.byref(d_clip), clip.nbytes)
    h.hipMalloc(C.byref(d_acc), fb)
    h.hipMemcpy(d_clip, clip.ctypes.data, clip.nbytes, 1)
    L.mdh_max_stack(0, d_clip, K, fb, 1, d_acc, 1)
    t0 = time.perf_counter()
    for _ in range(20):
        L.mdh_max_stack(0, d_clip, K, fb, 1, d_acc, 1)
    dt = (time.perf_counter() - t0) / 20
    acc = np.empty((1080, 1920, 3), np.uint8)
    h.hipMemcpy(acc.ctypes.data, d_acc, fb, 2)
    t1 = time.perf_counter()
    want = clip.max(axis=0)
    cpu_dt = time.perf_counter() - t1
    out["max_stack"] = {"frames": K, "frame": [1920, 1080, 3], "ms": dt * 1e3, "GBps": clip.nbytes / dt / 1e9,
                        "frames_per_s": K / dt, "parity": bool(np.array_equal(acc, want)), "cpu_numpy_ms": cpu_dt * 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
