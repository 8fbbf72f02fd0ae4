"""Host-side helpers of the line-detector path (NumPy only)."""
from __future__ import annotations

import datetime

import numpy as np

from . import _lib

VERSION = "V2.4.0"  # the MDRF / config format version this build is interchangeable with (utils.py:20)
BUILD = "mi355x"
PI = np.pi / 180.0

# global/class_name.txt of the reference plus the two generated labels.  NOTE the reference's inconsistency
# (SURVEY 0.6): utils.py:43-46 names id 8 "DROPPED" and id 9 "OTHERS", while collector.Name2Label says
# OTHERS() == 8 and DROPPED() == 9.  Both are reproduced literally because MDRF output depends on them.
CLASS_NAMES = ["METEOR", "PLANE/SATELLITE", "RED_SPRITE", "LIGHTNING", "JET", "RARE_SPRITE", "SPACECRAFT", "BUGS"]
ID2NAME = {i: n for i, n in enumerate(CLASS_NAMES)}
ID2NAME[len(CLASS_NAMES)] = "DROPPED"
ID2NAME[len(CLASS_NAMES) + 1] = "OTHERS"
NAME2ID = {n: i for i, n in ID2NAME.items()}
NUM_CLASS = len(ID2NAME)


def get_num_class() -> int:
    return NUM_CLASS


def select_std_roi(mask: np.ndarray, area: float):
    """Sub-window for the noise estimate: centred, area fraction `area`, moved up in 10-row steps while that
    increases the unmasked share (reference: Detector.py:93-122).  Returns (row0, col0, row1, col1)."""
    h, w = mask.shape[:2]
    if area <= 0:
        raise ValueError("binary.area must be > 0 (the reference cannot build its sub-window for area == 0)")
    side = area**0.5
    sh, sw = int(h * side), int(w * side)
    if sh <= 0 or sw <= 0:
        raise ValueError("binary.area too small for this frame size")
    top, left = (h - sh) // 2, (w - sw) // 2
    total = sh * sw
    share = np.sum(mask[top:top + sh, left:left + sw]) / total
    while share < 1:
        top -= 10
        moved = np.sum(mask[top:top + sh, left:left + sw]) / total
        if moved < share or top < 0:
            top += 10
            break
        share = moved
    return (top, left, top + sh, left + sw)


def lineset_nms(lines):
    """Line-set NMS (reference: utils.py:780-839), run by the native library on the host."""
    return _lib.lineset_nms(lines)


def frame2ts(frame: int, fps: float) -> str:
    stamp = datetime.datetime.fromtimestamp(frame / fps, tz=datetime.timezone.utc)
    return stamp.strftime("%H:%M:%S.%f")[:-3]


def ts2frame(time: str, fps: float) -> int:
    assert time.count(":") == 2, f"Invaild time string: \":\" in \"{time}\" should appear exactly 2 times."
    fmt = "%H:%M:%S.%f" if "." in time else "%H:%M:%S"
    dt = datetime.datetime.strptime(time, fmt)
    sec = dt.hour * 3600 + dt.minute * 60 + dt.second + dt.microsecond / 1e6
    return int(round(sec * fps))


def time2frame(time_ms: int, fps: float) -> int:
    return int(round(time_ms / 1000 * fps))


def frame2time(frame: int, fps: float) -> int:
    return int(round(frame * 1000 / fps))


def pt_len_sqr(p1, p2):
    if isinstance(p1, np.ndarray) and isinstance(p2, np.ndarray):
        return (p1[..., 1] - p2[..., 1])**2 + (p1[..., 0] - p2[..., 0])**2
    return (p1[1] - p2[1])**2 + (p1[0] - p2[0])**2


def pt_len(p1, p2):
    return np.sqrt(pt_len_sqr(p1, p2))


def pt_drct(p1, p2) -> float:
    return np.arccos((p2[1] - p1[1]) / pt_len(p1, p2))
