#!/usr/bin/env python3
"""The reference's own protocol, one frame at a time: detector.update(frame); detector.detect() (MetDetPy.py:197-198), frames in
pinned host memory.  Prints the rate, the host-side split (update / detect) and the per-stage GPU spans of a frame's pass."""
import ctypes
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from metdetpy_amd import M3Detector
from metdetpy_amd.synth import SkyStream


def main():
    H, W, n = 1080, 1920, 8
    frames = SkyStream(H, W, bench.WORKLOADS["C2"]["seed"]).take(250)
    det = M3Detector(1, n, np.ones((H, W), np.uint8), 10, bench.make_cfg(), None, max_batch=8)
    ctx = det._ctx
    pool = ctx.host_alloc(frames.nbytes)
    ctypes.memmove(pool, frames.ctypes.data, frames.nbytes)
    pinned = np.ctypeslib.as_array((ctypes.c_uint8 * frames.nbytes).from_address(pool)).reshape(frames.shape)
    for f in pinned[:40]:
        det.update(f)
        det.detect()
    out = {}
    for label, prof in (("plain", False), ("with_stage_events", True)):
        if prof:
            ctx.profile_enable(1)
            ctx.profile_read(True)
        tu = td = 0.0
        segs = 0
        N = int(os.environ.get("PF_N", "3000"))
        busy_t = quiet_t = 0.0  # frames whose Hough stage had something to replay (raw segments came out) / the others
        busy_n = 0
        t0 = time.perf_counter()
        for k in range(N):
            f = pinned[k % len(pinned)]
            a = time.perf_counter()
            det.update(f)
            b = time.perf_counter()
            lines, _ = det.detect()
            e = time.perf_counter()
            td += e - b
            tu += b - a
            segs += len(lines)
            if det.lines_num:
                busy_t += e - a
                busy_n += 1
            else:
                quiet_t += e - a
        dt = time.perf_counter() - t0
        out[label] = {"frames_per_s": N / dt, "us_per_frame": dt / N * 1e6, "update_us": tu / N * 1e6, "detect_us": td / N * 1e6, "segments": segs,
                      "frames_with_segments": busy_n, "us_per_frame_with_segments": busy_t / max(busy_n, 1) * 1e6,
                      "us_per_frame_without_segments": quiet_t / max(N - busy_n, 1) * 1e6}
        if prof:
            out[label]["gpu_span_us"] = {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in ctx.profile_read(True).items() if v[1]}
            ctx.profile_enable(False)
    print(json.dumps({"protocol": "update(frame) / detect() per frame, frames in pinned host memory", "frame": [W, H], **out}))
    del pinned
    ctx.host_free(pool)
    det.close()


if __name__ == "__main__":
    main()
