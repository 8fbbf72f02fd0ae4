#!/usr/bin/env python3
"""How long do the host's own calls take in the pipelined loop?  bench.py's stream loop with the duration of every mdh_submit_batch
and mdh_collect_batch call: a pass whose submit + collect cost the host more than the GPU needs for it is bound by the host.
    python tools/host_call_times.py --workload C1s [--window 25] [--batch 250] [--steps 40]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metdetpy_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="C2")
ap.add_argument("--window", type=int, default=None)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--unique", type=int, default=250)
ap.add_argument("--batch", type=int, default=None, help="frames per pass (default: the workload's)")
a = ap.parse_args()
wl = dict(bench.WORKLOADS[a.workload])
if a.window:
    wl["n"] = a.window
B, H, W, n = a.batch or wl["batch"], wl["H"], wl["W"], wl["n"]
mask = np.ones((H, W), np.uint8)
if a.workload == "C3" and bench.load_mask(H, W) is not None:
    mask = bench.load_mask(H, W)
frames = bench.make_stream_frames(wl, 0, mask, B, a.unique)
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


def run(nsteps):
    sub, col = [], []
    t0 = time.perf_counter()
    for s in range(nsteps):
        if ctx.in_flight >= ctx.max_in_flight:
            t = time.perf_counter(); ctx.collect_into(info, raw, nms, prob); col.append(time.perf_counter() - t)
        t = time.perf_counter(); ctx.submit(pool, B, True); sub.append(time.perf_counter() - t)
    while ctx.in_flight:
        t = time.perf_counter(); ctx.collect_into(info, raw, nms, prob); col.append(time.perf_counter() - t)
    return time.perf_counter() - t0, np.array(sub) * 1e3, np.array(col) * 1e3


run(8)
dt, sub, col = run(a.steps)
print(f"{a.workload} n={n} B={B}: {dt / a.steps * 1e3:.3f} ms per pass = {B * a.steps / dt:,.0f} frames/s; submit call: median {np.median(sub):.3f} ms, max {sub.max():.3f}; "
      f"collect call: median {np.median(col):.3f} ms (waits for the GPU when it is ahead), min {col.min():.3f}, p10 {np.percentile(col, 10):.3f}")
