"""Exposure estimation: how many video frames one real exposure spans (cameras recording with a shutter slower than the
frame rate repeat each exposure), and the `exp_time` / `exp_frame` the loader derives from it.

Mirrors `rf_estimator` / `_rf_est_kernel` (videoloader.py:796-870), `sigma_clip` (utils.py:516-540) and
`VanillaVideoLoader.init_exp_time` (videoloader.py:440-494).  The only pixel work is the per-frame sum, which runs on the
GPU (`mdh_frame_sums`); everything after it is a few hundred scalars on the host, restated operation for operation and
pinned to the reference's own outputs (tests/golden/g8_exposure.json).
"""
from __future__ import annotations

from typing import Callable, Union

import numpy as np

from . import _lib

SHORT_LENGTH_THRESHOLD = 300  # videoloader.py:39
RF_ESTIMATE_LENGTH = 100      # videoloader.py:40
SLOW_EXP_TIME = 1 / 4         # videoloader.py:41
MAX_LOOP_CNT = 10             # utils.py:25


def frame_sums(frames, K=None, frame_bytes=None, on_device=False, device: int = 0) -> np.ndarray:
    """np.sum(frame) for each of K frames, on the GPU.  frames: (K, ...) uint8 array, or a device pointer with K and
    frame_bytes.  Returns float64 (what the reference stores the sums as; exact below 2^53)."""
    L = _lib.load()
    if on_device:
        ptr = int(frames)
    else:
        frames = np.ascontiguousarray(frames, dtype=np.uint8)
        K = len(frames)
        frame_bytes = frames[0].nbytes
        if frame_bytes % 16:  # pad each frame with zeros up to a multiple of 16 bytes
            pad = 16 - frame_bytes % 16
            frames = np.concatenate([frames.reshape(K, -1), np.zeros((K, pad), np.uint8)], axis=1)
            frame_bytes += pad
        ptr = frames.ctypes.data
    out = np.zeros(int(K), np.uint64)
    _lib.check(L.mdh_frame_sums(int(device), ptr, int(K), int(frame_bytes), int(bool(on_device)), out.ctypes.data),
               "mdh_frame_sums")
    return out.astype(np.float64)


def sigma_clip(sequence, sigma=3.00) -> np.ndarray:
    """utils.py:516-540."""
    seq = np.array(sequence).reshape((-1, ))
    mean, std = np.mean(seq), np.std(seq)
    loops = 0
    while True:
        seq = seq[np.abs(mean - seq) <= sigma * std]
        new_mean, new_std = np.mean(seq), np.std(seq)
        if new_mean == mean:
            return seq
        mean, std = new_mean, new_std
        loops += 1
        if loops >= MAX_LOOP_CNT:
            return seq


def rf_intervals(f_sum: np.ndarray) -> np.ndarray:
    """Distances between the frames where the brightness series turns (second difference changes sign and the first
    difference is not flat): _rf_est_kernel, videoloader.py:809-819."""
    f_sum = np.asarray(f_sum, dtype=float)
    a0, a1, a2, a3 = f_sum[:-3], f_sum[1:-2], f_sum[2:-1], f_sum[3:]
    diff = f_sum[1:] - f_sum[:-1]
    pos = np.where((2 * a2 - (a1 + a3) > 0) & (2 * a1 - (a0 + a2) < 0) & (np.abs(diff[1:-1]) > 0.01))[0]
    return pos[1:] - pos[:-1]


def estimation_windows(start_frame: int, end_frame: int):
    """The frame ranges rf_estimator samples: everything for short videos, else 100 frames at the start, the middle and
    the end (videoloader.py:840-858)."""
    total = end_frame - start_frame
    if total < SHORT_LENGTH_THRESHOLD:
        return [(start_frame, end_frame)]
    return [(start_frame, start_frame + RF_ESTIMATE_LENGTH),
            (start_frame + (total - RF_ESTIMATE_LENGTH) // 2, start_frame + (total + RF_ESTIMATE_LENGTH) // 2),
            (end_frame - RF_ESTIMATE_LENGTH, end_frame)]


def rf_estimator(sums_of: Callable[[int, int], np.ndarray], start_frame: int, end_frame: int) -> Union[float, int]:
    """rf_estimator (videoloader.py:822-870).  sums_of(lo, hi) returns the per-frame sums of frames [lo, hi) read with
    exp_frame = 1 (e.g. `lambda lo, hi: frame_sums(read(lo, hi))`)."""
    parts = [rf_intervals(sums_of(lo, hi)) for lo, hi in estimation_windows(start_frame, end_frame)]
    intervals = parts[0] if len(parts) == 1 else np.concatenate(parts)
    if len(intervals) == 0:
        return 1
    return np.round(np.min([np.median(intervals), np.mean(sigma_clip(intervals))]))


def init_exp_time(exp_option, fps: float, upper_bound: float, estimate: Callable[[], float], logger=None) -> float:
    """VanillaVideoLoader.init_exp_time (videoloader.py:440-494); `estimate()` is rf_estimator for this video."""
    warn = getattr(logger, "warning", None) or (lambda *_: None)
    assert isinstance(exp_option, (str, float, int)), f"exp_option should be either <str, float, int>, got {type(exp_option)}."
    if fps <= int(1 / upper_bound):
        warn(f"Slow FPS detected. Use {1/fps:.2f}s directly.")
        return 1 / fps
    if isinstance(exp_option, str):
        if exp_option == "real-time":
            return 1 / fps
        if exp_option == "slow":
            return SLOW_EXP_TIME
        if exp_option == "auto":
            rf = estimate()
            if rf / fps >= upper_bound:
                warn(f"Unexpected exposuring time (too long):{rf/fps:.2f}s. Use {upper_bound:.2f}s instead.")
            return min(rf / fps, upper_bound)
        try:
            exp_time = float(exp_option)
        except ValueError as e:
            raise ValueError(f"{e.__repr__()}: Invalid exp_time string value: It should be selected from [float], [int], "
                             f"real-time\",\"auto\" and \"slow\", got {exp_option}.")
    else:
        exp_time = exp_option
    if exp_time * fps < 1:
        warn(f"Invalid exposuring time (too short). Use {1/fps:.2f}s instead.")
        return 1 / fps
    return float(exp_time)


def exposure_params(exp_time: float, fps: float):
    """exp_frame, eq_fps, eq_int_fps as the loader derives them (videoloader.py:186-192, 313)."""
    import math
    return dict(exp_frame=int(round(exp_time * fps)), eq_fps=1 / exp_time, eq_int_fps=math.floor(1 / exp_time))
