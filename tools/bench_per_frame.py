#!/usr/bin/env python3
"""The reference's own protocol, one frame at a time from host memory: detector.update(frame); detector.detect()
(MetDetPy.py:197-198).  Every call is a host round trip (H2D copy of the frame, ~10 launches, D2H of the results)."""
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
    frames = SkyStream(H, W, bench.WORKLOADS["C2"]["seed"]).take(300)
    det = M3Detector(1, n, np.ones((H, W), np.uint8), 10, bench.make_cfg(), None, max_batch=8)
    for f in frames[:40]:
        det.update(f)
        det.detect()
    t0 = time.perf_counter()
    segs = 0
    for f in frames[40:]:
        det.update(f)
        lines, _ = det.detect()
        segs += len(lines)
    dt = time.perf_counter() - t0
    print(json.dumps({"protocol": "update(frame) / detect() per frame, host frames", "frame": [W, H], "frames_per_s": (len(frames) - 40) / dt,
                      "ms_per_frame": dt / (len(frames) - 40) * 1e3, "segments": segs}))
    det.close()


if __name__ == "__main__":
    main()
