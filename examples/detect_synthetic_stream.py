#!/usr/bin/env python3
"""End to end on a synthetic night-sky stream: frames -> M3Detector (batched HIP passes) -> native MeteorCollector -> MDRF
document on stdout.  Needs an MI355X.

    python examples/detect_synthetic_stream.py [--frames 3000] [--size 1920x1080] [--fps 25]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metdetpy_amd import BinaryCfg, BinaryCoreCfg, CollectorCfg, DynamicCfg, HoughLineCfg, MeteorCfg, RuntimeParams
from metdetpy_amd.metstruct import RecheckSwitch
from metdetpy_amd.pipeline import detect_stream, to_mdrf
from metdetpy_amd.synth import SkyStream


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--fps", type=float, default=25.0)
    ap.add_argument("--batch", type=int, default=512)
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.split("x"))
    # config/m3det_normal.json of the reference: detector and collector blocks (re-check off)
    det_cfg = BinaryCfg(BinaryCoreCfg(True, 7, "normal", 0.1, 2), HoughLineCfg(10, 10, 10), DynamicCfg(True, 5))
    col_cfg = CollectorCfg(MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})
    rt = RuntimeParams(fps=a.fps, exp_frame=1, eq_fps=a.fps, eq_int_fps=int(a.fps), exp_time=1 / a.fps, runtime_size=[W, H],
                       raw_size=[W, H], positive_category_list=["METEOR", "RED_SPRITE"])
    stream = SkyStream(H, W, 20220413, speed=0.6 * W / a.fps / 4, streak_period=int(6 * a.fps), streak_life=int(0.8 * a.fps))
    mask = np.ones((H, W), np.uint8)

    def frames():
        for _ in range(a.frames):
            yield stream.next()

    t0 = time.perf_counter()
    records, info = detect_stream(frames(), mask, det_cfg, col_cfg, rt, window_sec=1.0, batch=a.batch)
    dt = time.perf_counter() - t0
    from metdetpy_amd.metstruct import BasicInfo
    doc = to_mdrf(records, BasicInfo("synthetic", "SkyStream", None, 0, int(a.frames / rt.fps * 1000), [W, H], [W, H], rt.exp_time,
                                     a.frames, rt.fps, "synthetic sky, metdetpy_amd")).to_dict()
    print(json.dumps(doc, indent=1))
    print(f"{a.frames} frames of {W}x{H} in {dt:.1f} s (frame synthesis on the host included), {len(records)} records, "
          f"{int(info['n_raw'].sum())} raw segments", file=sys.stderr)


if __name__ == "__main__":
    main()
