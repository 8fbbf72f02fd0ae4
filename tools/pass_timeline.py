#!/usr/bin/env python3
"""When do the passes of a short pipelined run complete?  bench.py's stream loop with a host timestamp at every submit and
collect: shows the pipeline's fill and drain next to its steady pace (why --steps 20 reads lower than a 2 s run)."""
import argparse
import sys
import os
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metdetpy_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--batch", type=int, default=None)
a = ap.parse_args()
wl = bench.WORKLOADS[a.workload]
B = a.batch or wl["batch"]
H, W, n = wl["H"], wl["W"], wl["n"]
mask = np.ones((H, W), np.uint8)
if a.workload == "C3" and bench.load_mask(H, W) is not None:
    mask = bench.load_mask(H, W)
frames = bench.make_stream_frames(wl, 0, mask, B, 250)
det = bench.make_detector(n, mask, 0, B)
ctx = det._ctx
pool = ctx.device_alloc(B * H * W)
for start in range(0, B, len(frames)):
    cnt = min(len(frames), B - start)
    ctx.device_upload(pool + start * H * W, frames[:cnt])
info = np.zeros(B, _lib.FRAME_INFO_DTYPE)
raw = np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32)
nms = np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32)
prob = np.zeros((B, _lib.MDH_MAX_LINES), np.float64)


def run(nsteps, log=None):
    t0 = time.perf_counter()
    for s in range(nsteps):
        if ctx.in_flight >= ctx.max_in_flight:
            ctx.collect_into(info, raw, nms, prob)
            if log is not None:
                log.append(("collect", time.perf_counter() - t0))
        ctx.submit(pool, B, True)
        if log is not None:
            log.append(("submit", time.perf_counter() - t0))
    while ctx.in_flight:
        ctx.collect_into(info, raw, nms, prob)
        if log is not None:
            log.append(("collect", time.perf_counter() - t0))
    return time.perf_counter() - t0


run(a.warmup)
for rep in range(3):
    log = []
    dt = run(a.steps, log)
    col = [t for k, t in log if k == "collect"]
    sub = [t for k, t in log if k == "submit"]
    d = np.diff(col) * 1e3
    print(f"rep {rep}: {a.steps} passes in {dt * 1e3:.2f} ms = {dt / a.steps * 1e3:.3f} ms/pass; first collect at {col[0] * 1e3:.2f} ms; "
          f"first 6 submits done at {sub[min(5, len(sub) - 1)] * 1e3:.2f} ms; collect intervals (ms): " + " ".join(f"{x:.2f}" for x in d))
