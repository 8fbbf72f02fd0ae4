#!/usr/bin/env python3
"""Batched loader preprocessing with everything resident in HBM: n_out output frames per launch.
C1: 3840x2160x3 BGR -> 960x540 gray (resize INTER_LINEAR, gray, mask), exp_frame = 1 and 3."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.imgproc import Preprocessor
from bench import make_cfg


def main():
    sh, sw, dw, dh = 2160, 3840, 960, 540
    rng = np.random.default_rng(1)
    base = rng.integers(0, 256, (4, sh, sw, 3), dtype=np.uint8)
    mask = np.ones((dh, dw), np.uint8)
    det = M3Detector(1, 8, mask, 10, make_cfg(), None, max_batch=8)  # only for the device helpers
    ctx = det._ctx
    out = {}
    for exp, n_out in ((1, 64), (3, 32)):
        nsrc = exp * n_out
        d_src = ctx.device_alloc(nsrc * base[0].nbytes)
        for i in range(nsrc):
            ctx.device_upload(d_src + i * base[0].nbytes, base[i % 4])
        d_out = ctx.device_alloc(n_out * dh * dw)
        pre = Preprocessor((sh, sw), (dw, dh), mask)
        for _ in range(3):
            pre.batch_device(d_src, n_out, exp, d_out)
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            pre.batch_device(d_src, n_out, exp, d_out)
        dt = (time.perf_counter() - t0) / reps
        src_bytes = nsrc * base[0].nbytes
        out[f"exp{exp}"] = {"n_out": n_out, "ms_per_launch": dt * 1e3, "output_frames_per_s": n_out / dt,
                            "source_frames_per_s": nsrc / dt, "source_GBps_nominal": src_bytes / dt / 1e9,
                            "rows_touched_GBps": src_bytes / 2 / dt / 1e9}
        pre.close()
        ctx.device_free(d_src)
        ctx.device_free(d_out)
    det.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
