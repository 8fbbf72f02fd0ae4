"""CPU oracle of `hough_mode = accumulator` (MDH_HOUGH_ACCUMULATOR, include/metdet_hip.h) -- TEST INFRASTRUCTURE ONLY.

This mode is NOT `cv2.HoughLinesP` (the reference's call, Detector.py:347-353, is replayed exactly by the default mode, whose
oracle is m3_oracle.hough_lines_p).  It is the order-free full-accumulator transform BASELINE.json's north_star asks for next to
it ("Hough accumulator ... LDS-binned atomics", rho / theta within +-1 px / +-0.5 deg of OpenCV's), and its output is a function
of the binary image alone.  This file is its SPECIFICATION, stated independently of the kernels; the HIP path must equal it bit for
bit (tests/test_gpu_hough_accumulator.py), and the tolerance tests tie it to cv2.HoughLinesP's restatement and to ground truth.

Everything the transform shares with OpenCV's progressive transform is taken from the same place (hough.cpp
HoughLinesProbabilistic): theta = (float)(pi/180), 180 angles, trig table (float)(cos((double)n * theta)), rho =
cvRound(x * cos + y * sin) in float32 without contraction, the 16.16 fixed-point line walk with `lineGap`, the `good` test
against `lineLength`, segments as [x_end0, y_end0, x_end1, y_end1].  What differs:

  1. VOTE: every lit pixel votes every angle once:  acc[n][rho_n(x, y)] += 1.
  2. PEAK: while the largest cell holds >= threshold votes: take the largest cell; among equal counts the SMALLER ANGLE wins, then
     the SMALLER RHO ("first maximum" of a scan over n, then rho).
  3. SEED: of the still-lit pixels that vote for that cell (row-major order) the MIDDLE one, voters[len(voters) // 2].
  4. LINE: OpenCV's walk from the seed along angle n in both directions (gap rule on the CURRENT mask), ends, `good`; every still-lit
     pixel on the walk up to the two ends is cleared AND its votes are taken back -- whether or not the segment is good (the
     accumulator always equals the votes of the pixels still lit, so the loop ends after at most N peaks); a good segment is emitted.
  5. Frames with more than 8192 lit pixels are handed to the exact progressive transform (m3_oracle.hough_lines_p) in both modes,
     and frames with fewer lit pixels than `threshold` have no segment.

`hough_lines_acc` is the NumPy statement (whole-accumulator arg-max per peak: small images); `c_hough_lines_acc` binds the C port
in metdet_oracle.c (per-angle cached maxima: full-size images), cross-checked against each other in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import m3_oracle as mo

MAX_LIST = 8192  # HOUGH_MAX_LIST: above it both modes replay cv2.HoughLinesP exactly


def hough_lines_acc(image: np.ndarray, threshold: int, min_line_length: float, max_line_gap: float) -> np.ndarray:
    """(N, 4) int32 segments [x1, y1, x2, y2], strongest peak first."""
    assert image.dtype == np.uint8 and image.ndim == 2
    height, width = image.shape
    ys, xs = np.nonzero(image)  # row-major
    N = len(xs)
    if N < threshold:
        return np.zeros((0, 4), np.int32)
    if N > MAX_LIST:
        return mo.hough_lines_p(image, 1, np.pi / 180.0, threshold, min_line_length, max_line_gap)
    theta32 = np.float32(np.pi / 180.0)
    line_length = mo.cv_round(min_line_length)
    line_gap = mo.cv_round(max_line_gap)
    numangle = 180
    numrho = (width + height) * 2 + 1
    half = (numrho - 1) // 2
    tcos, tsin = mo.hough_trig_table(theta32, numangle, np.float32(1))
    ang = np.arange(numangle)

    def rho_all(x, y):  # (len(x), 180) rho bins, float32 mul, mul, add -- not fused; cvRound
        r = x.astype(np.float32)[:, None] * tcos[None, :] + y.astype(np.float32)[:, None] * tsin[None, :]
        return np.rint(r).astype(np.int64) + half

    acc = np.zeros((numangle, numrho), np.int32)
    R = rho_all(xs, ys)
    for n in range(numangle):
        acc[n] += np.bincount(R[:, n], minlength=numrho).astype(np.int32)
    mask = (image != 0)
    alive = np.ones(N, bool)  # list entry still lit
    index_of = -np.ones((height, width), np.int64)
    index_of[ys, xs] = np.arange(N)
    out = []
    shift = 16
    for _ in range(N + 1):
        row_max = acc.max(axis=1)
        bn = int(np.argmax(row_max))  # first maximum: smaller angle
        count = int(row_max[bn])
        if count < threshold:
            break
        rho = int(np.argmax(acc[bn]))  # first maximum: smaller rho
        voters = np.nonzero(alive & (R[:, bn] == rho))[0]
        assert len(voters) == count
        seed = int(voters[len(voters) // 2])
        j, i = int(xs[seed]), int(ys[seed])
        a = np.float32(-tsin[bn])
        b = np.float32(tcos[bn])
        x0, y0 = j, i
        if abs(a) > abs(b):
            xflag = True
            dx0 = 1 if a > 0 else -1
            dy0 = mo.cv_round(b * np.float32(1 << shift) / np.abs(a))
            y0 = (y0 << shift) + (1 << (shift - 1))
        else:
            xflag = False
            dy0 = 1 if b > 0 else -1
            dx0 = mo.cv_round(a * np.float32(1 << shift) / np.abs(b))
            x0 = (x0 << shift) + (1 << (shift - 1))
        ends = [[0, 0], [0, 0]]
        for k in range(2):
            gap = 0
            x, y = x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> shift) if xflag else (x >> shift, y)
                if j1 < 0 or j1 >= width or i1 < 0 or i1 >= height:
                    break
                if mask[i1, j1]:
                    gap = 0
                    ends[k] = [j1, i1]
                else:
                    gap += 1
                    if gap > line_gap:
                        break
                x += dx
                y += dy
        good = abs(ends[1][0] - ends[0][0]) >= line_length or abs(ends[1][1] - ends[0][1]) >= line_length
        for k in range(2):
            x, y = x0, y0
            dx, dy = (dx0, dy0) if k == 0 else (-dx0, -dy0)
            while True:
                j1, i1 = (x, y >> shift) if xflag else (x >> shift, y)
                if mask[i1, j1]:
                    p = int(index_of[i1, j1])
                    acc[ang, R[p]] -= 1  # taken back whether or not the segment is good
                    alive[p] = False
                    mask[i1, j1] = False
                if i1 == ends[k][1] and j1 == ends[k][0]:
                    break
                x += dx
                y += dy
        if good:
            out.append([ends[0][0], ends[0][1], ends[1][0], ends[1][1]])
    return np.array(out, dtype=np.int32).reshape(-1, 4)


def c_hough_lines_acc(image, threshold, min_line_length, max_line_gap, cap: int = 1 << 16) -> np.ndarray:
    """The C port (oracle/metdet_oracle.c: oacc_hough_lines): same contract."""
    from . import c_oracle as co
    L = co.lib()
    if not hasattr(L, "_acc_bound"):
        L.oacc_hough_lines.argtypes = [co.u8p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, co.i32p, C.c_int]
        L.oacc_hough_lines.restype = C.c_int
        L._acc_bound = True
    img = np.ascontiguousarray(image)
    buf = np.zeros((cap, 4), np.int32)
    n = L.oacc_hough_lines(img, img.shape[0], img.shape[1], int(threshold), float(min_line_length), float(max_line_gap), buf, cap)
    assert 0 <= n <= cap
    return buf[:n].copy()


def adaptive_gap(image, mask_area: float, max_gap: float) -> float:
    """Detector.py:342-344: the maxLineGap M3Detector hands to the transform."""
    dst_sum = float(np.sum(image / 255.) / mask_area * 100)
    return max(0, 1 - dst_sum / 0.05) * max_gap
