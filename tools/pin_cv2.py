#!/usr/bin/env python3
"""Pin the OpenCV side of the hot path.  Run this ONCE on any machine where `import cv2` works (opencv-python >= 4.9, the
reference's requirement; never needed on the GPU box):

    python tools/pin_cv2.py                                  # the cv2 operators on the seeded inputs of tests/cv2_cases.py
    python tools/pin_cv2.py --reference /path/to/MetDetPy    # ... and the reference's own M3Detector / ClassicDetector

It writes tests/golden/g9_cv2_{stencil,hough,color,classic,m3}.npz + g9_cv2_meta.json (inputs are regenerated from seeds by the tests,
only what cv2 / the reference RETURNED is stored).  With those files present, tests/test_cv2_pinned.py stops skipping with
"PARITY UNPINNED" and checks the CPU oracle -- and, under -m gpu, the HIP path -- against them bit for bit:

    cv2.medianBlur / threshold / morphologyEx(MORPH_CLOSE) / erode / dilate / absdiff      Detector.py:184, 241, 266-288, 329-335
    cv2.HoughLinesP(img, 1, pi/180, thr, minLineLength, maxLineGap)                        Detector.py:289-293, 347-353
    cv2.cvtColor(BGR2GRAY), cv2.resize(INTER_LINEAR)                                       imgproc.py:82-88, fileio.py:250-292
    ClassicDetector.detect's cv2 call sequence on four-frame windows (dst, lines)           Detector.py:257-296
    the two contested spots: BGR2GRAY on pixels where the 15-bit and the 14-bit coefficient sets differ; 4:1 INTER_LINEAR
    HoughLinesP with maxLineGap at .5 (cvRound: half to even)                               Detector.py:343-353
    M3Detector.update / detect series (threshold, dst, lines, cls_pred)                     Detector.py:302-392

`generate(outdir, cv2)` takes the module as an argument so that the file formats can be exercised without OpenCV (the tests feed
it a stand-in built from the oracle and check the round trip)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cv2_cases as cases  # noqa: E402


def pin_stencil(cv2):
    out = {}
    k3 = cv2.getStructuringElement(cv2.MORPH_RECT, (3, 3))
    prev = None
    for name, img in cases.stencil_inputs():
        if name.startswith("gray"):
            out[f"{name}/median3"] = cv2.medianBlur(img, 3)
            for t in cases.THRESHOLDS:
                out[f"{name}/thr{t}"] = cv2.threshold(img, t, 255, cv2.THRESH_BINARY)[-1]
            if prev is not None and prev.shape == img.shape:
                out[f"{name}/absdiff_prev"] = cv2.absdiff(prev, img)
            prev = img
        else:
            out[f"{name}/close3"] = cv2.morphologyEx(img, cv2.MORPH_CLOSE, k3)
            out[f"{name}/erode3"] = cv2.erode(img, k3)
            out[f"{name}/dilate3"] = cv2.dilate(img, k3)
            out[f"{name}/median3"] = cv2.medianBlur(img, 3)
    return out


def pin_hough(cv2):
    out = {}
    for name, img in cases.hough_inputs():
        for (thr, min_len, gap) in cases.HOUGH_PARAMS:
            res = cv2.HoughLinesP(img, 1, np.pi / 180.0, thr, minLineLength=min_len, maxLineGap=gap)
            arr = np.zeros((0, 4), np.int32) if res is None else np.asarray(res, np.int32).reshape(-1, 4)
            out[f"{name}/t{thr}_l{min_len}_g{gap}"] = arr
    return out


def pin_color(cv2):
    out = {}
    for name, img in cases.color_inputs():
        out[f"{name}/gray"] = cv2.cvtColor(img, cv2.COLOR_BGR2GRAY)
        for (w, h) in cases.RESIZES[img.shape[:2]]:
            out[f"{name}/resize_{w}x{h}"] = cv2.resize(img, (w, h), interpolation=cv2.INTER_LINEAR)
            out[f"{name}/gray_of_resize_{w}x{h}"] = cv2.cvtColor(out[f"{name}/resize_{w}x{h}"], cv2.COLOR_BGR2GRAY)  # the loader's order
            out[f"{name}/resize_gray_{w}x{h}"] = cv2.resize(out[f"{name}/gray"], (w, h), interpolation=cv2.INTER_LINEAR)
    return out


def pin_classic(cv2):
    """ClassicDetector.detect's own sequence of cv2 calls (Detector.py:257-296), frame by frame on the seeded frames: the
    reference's class is not needed for it, only cv2."""
    out = {}
    k3 = cv2.getStructuringElement(cv2.MORPH_RECT, (3, 3))
    frames = cases.classic_frames()
    thr_h, min_len, gap = cases.CLASSIC["hough"]
    for T in cases.CLASSIC["thresholds"]:
        for t in range(3, len(frames)):
            f0, f1, f2, f3 = frames[t], frames[t - 1], frames[t - 2], frames[t - 3]  # f0 newest
            _, d = cv2.threshold(cv2.absdiff(f1, f0), T, 255, cv2.THRESH_BINARY)
            d = 255 - cv2.dilate(d, k3)
            a, b = cv2.bitwise_and(d, f3), cv2.bitwise_and(d, f2)
            _, dst = cv2.threshold(cv2.absdiff(a, b), T, 255, cv2.THRESH_BINARY)
            dst = cv2.dilate(dst, k3)
            res = cv2.HoughLinesP(dst, 1, np.pi / 180.0, thr_h, minLineLength=min_len, maxLineGap=gap)
            out[f"T{T}/{t}/dst"] = np.packbits(dst > 0, axis=1)
            out[f"T{T}/{t}/lines"] = np.zeros((0, 4), np.int32) if res is None else np.asarray(res, np.int32).reshape(-1, 4)
    return out


def pin_m3(reference):
    """The reference's own detectors on the seeded synthetic sky (needs the reference's full environment)."""
    sys.path.insert(0, reference)
    from MetLib.Detector import ClassicDetector, M3Detector
    from MetLib.metstruct import BinaryCfg
    from MetLib.utils import get_num_class
    from metdetpy_amd.synth import SkyStream
    out = {}
    for S in (cases.M3_STREAM, cases.M3_STREAM_LONG):
        _pin_m3_stream(S, out, M3Detector, ClassicDetector, BinaryCfg, get_num_class, SkyStream)
    return out


def _pin_m3_stream(S, out, M3Detector, ClassicDetector, BinaryCfg, get_num_class, SkyStream):
    prefix = S.get("prefix", "")
    mask = np.ones((S["H"], S["W"]), np.uint8)
    for sens in S["sensitivities"]:
        cfg = BinaryCfg.from_dict({"binary": {"adaptive_bi_thre": True, "init_value": 7, "sensitivity": sens, "area": 0.1, "interval": 2},
                                   "hough_line": {"threshold": 10, "min_len": 10, "max_gap": 10}, "dynamic": {"dy_mask": True, "window_sec": 5}})
        for cls, tag in ((M3Detector, prefix + "m3"),) + (((ClassicDetector, "classic"),) if not prefix else ()):
            det = cls(window_sec=1, fps=S["n"], mask=mask, num_cls=get_num_class(), cfg=cfg, logger=None)
            stream = SkyStream(S["H"], S["W"], S["seed"], speed=S["speed"], streak_period=S["streak_period"])
            for t in range(S["frames"]):
                det.update(stream.next())
                lines, pred = det.detect()
                out[f"{tag}_{sens}/{t}/lines"] = np.asarray(lines, np.int32).reshape(-1, 4)
                out[f"{tag}_{sens}/{t}/cls_pred"] = np.asarray(pred, np.float64).reshape(len(out[f"{tag}_{sens}/{t}/lines"]), -1)
                out[f"{tag}_{sens}/{t}/scalars"] = np.array([det.bi_threshold, getattr(det, "lines_num", len(lines)),
                                                             getattr(det, "dst_sum", 0.0)], np.float64)
                if hasattr(det, "dst") and t % 5 == 0:
                    out[f"{tag}_{sens}/{t}/dst"] = np.packbits(np.asarray(det.dst) > 0, axis=1)
    return out


def generate(outdir, cv2, reference=None):
    os.makedirs(outdir, exist_ok=True)
    written = []
    for tag, fn in (("stencil", pin_stencil), ("hough", pin_hough), ("color", pin_color), ("classic", pin_classic)):
        np.savez_compressed(os.path.join(outdir, f"g9_cv2_{tag}.npz"), **fn(cv2))
        written.append(f"g9_cv2_{tag}.npz")
    if reference:
        np.savez_compressed(os.path.join(outdir, "g9_cv2_m3.npz"), **pin_m3(reference))
        written.append("g9_cv2_m3.npz")
    json.dump({"cv2_version": getattr(cv2, "__version__", "unknown"), "numpy_version": np.__version__, "files": written,
               "reference": bool(reference)}, open(os.path.join(outdir, "g9_cv2_meta.json"), "w"))
    return written


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--reference", default=None, help="checkout of LilacMeteorObservatory/MetDetPy (V2.4.0) with its requirements installed")
    a = ap.parse_args()
    import cv2
    print("opencv", cv2.__version__, "->", a.out)
    for f in generate(a.out, cv2, a.reference):
        print("  wrote", f)
    print("now run: python -m pytest tests/test_cv2_pinned.py   (and -m gpu on the MI355X box)")


if __name__ == "__main__":
    main()
