#!/usr/bin/env python3
"""Stress of the per-frame protocol's hand-over (the window kernel launched ahead of its frame): the same stream through a
detector without it (MDH_LEAN_MAIL=0) and through one with it forced on, the second with random pauses between the frames --
none, a busy wait around the kernel's patience (150 us: the expiry race), milliseconds -- and an occasional get_binary (cancel).
Every frame's threshold, lit count and segments must agree."""
import ctypes
import os
import random
import sys
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metdetpy_amd import M3Detector
from metdetpy_amd.synth import SkyStream

H, W, n = 1080, 1920, 8
N = int(os.environ.get("STRESS_N", "30000"))
frames = SkyStream(H, W, bench.WORKLOADS["C2"]["seed"]).take(250)


def run(mail, pauses):
    os.environ["MDH_LEAN_MAIL"] = "1" if mail else "0"
    os.environ["MDH_LEAN_MAIL_GAP"] = "1e12"
    os.environ["MDH_HOST_TRACE"] = "1"
    det = M3Detector(1, n, np.ones((H, W), np.uint8), 10, bench.make_cfg(), None, max_batch=8)
    ctx = det._ctx
    pool = ctx.host_alloc(frames.nbytes)
    ctypes.memmove(pool, frames.ctypes.data, frames.nbytes)
    pinned = np.ctypeslib.as_array((ctypes.c_uint8 * frames.nbytes).from_address(pool)).reshape(frames.shape)
    rng = random.Random(7)
    sig = []
    t0 = time.perf_counter()
    for k in range(N):
        if pauses:
            r = rng.random()
            if r < 0.05:
                t = time.perf_counter()
                while time.perf_counter() - t < rng.uniform(100e-6, 200e-6):
                    pass
            elif r < 0.07:
                time.sleep(0.002)
            elif r < 0.08:
                ctx.get_binary(-1, 0)
        det.update(pinned[k % 250])
        lines, _ = det.detect()
        info = ctx._info1[0]
        sig.append((int(info["bi_threshold"]), int(info["lit_count"]), int(info["n_raw"]), zlib.crc32(np.ascontiguousarray(det.linesp_ext).tobytes()),
                    zlib.crc32(np.ascontiguousarray(lines).tobytes())))
    dt = time.perf_counter() - t0
    del pinned
    ctx.host_free(pool)
    det.close()
    return sig, dt


a, ta = run(False, False)
b, tb = run(True, True)
bad = [k for k in range(N) if a[k] != b[k]]
print(f"{N} frames: without {N / ta:.0f} frames/s, with (pauses included) {N / tb:.0f} frames/s; mismatching frames: {len(bad)} {bad[:10]}")
sys.exit(1 if bad else 0)
