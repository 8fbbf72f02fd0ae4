"""Detector plugin classes with the reference's interface, backed by libmetdet_hip.so.

    BaseDetector(ABC) -> LineDetector -> {M3Detector, ClassicDetector}

Same constructor, `update(frame)`, `detect() -> (lines, cls_pred)` and attribute names as the reference
(Detector.py:130-157, 160-242, 245-299, 302-448), so `MetDetPy.detect_video` can select them by name
(`get_detector`, MetLib/__init__.py:34-46) and drive them unchanged.  All pixel work runs in hand-written
HIP kernels; this file only marshals arguments.  There is no CPU fallback.

Beyond the reference's per-frame protocol, `detect_batch(frames)` runs K update()+detect() pairs in one
native call (identical results), which is what the throughput numbers are measured on.
"""
from __future__ import annotations

from abc import ABCMeta, abstractmethod
from typing import Any, Optional

import numpy as np

from . import _lib
from .metstruct import BinaryCfg
from .utils import select_std_roi
from .visu import DrawRectVisu, ImgVisuAttrs, SquareColorPair, TextColorPair, TextVisu

NUM_LINES_TOOMUCH = 500
DEFAULT_INIT_VALUE = 5

class BaseDetector(metaclass=ABCMeta):
    """update(frame) / detect() / visu()  (Detector.py:130-157)."""

    @abstractmethod
    def __init__(self, *args: Any) -> None:
        pass

    @abstractmethod
    def update(self, new_frame) -> None:
        pass

    @abstractmethod
    def detect(self):
        pass

    def visu(self) -> list:
        return []


_EMPTY = np.array([])
_EMPTY.flags.writeable = False


class _StackView:
    """What callers read from `detector.stack` (SNR_SW): snr, std_roi, timer, n."""

    def __init__(self, n: int, std_roi):
        self.n = n
        self.std_roi = std_roi
        self.snr = 0
        self.timer = 0
        self.noise_cur_value = 0.0


class LineDetector(BaseDetector):
    """Binarise + Hough line detector base (Detector.py:160-242).  Input must be grayscale uint8."""

    sensitivity_func = {
        "low": lambda x: 2.0 * x**2 + 4.4,
        "normal": lambda x: 1.2 * x**2 + 3.6,
        "high": lambda x: 0.9 * x**2 + 3,
    }
    abs_sensitivity = {"high": 3, "normal": 5, "low": 7}
    _native_kind = _lib.MDH_DET_M3

    def __init__(self, window_sec: float, fps: float, mask: np.ndarray, num_cls: int, cfg: BinaryCfg, logger=None,
                 *, device: int = 0, max_batch: int = 64, hough_mode: str = "exact"):
        """hough_mode: "exact" (default) replays cv2.HoughLinesP segment for segment -- the parity path; "accumulator" runs the
        order-free full-accumulator transform (include/metdet_hip.h, MDH_HOUGH_ACCUMULATOR): different segments, same detections
        under the tolerance rule of SURVEY 8d."""
        if hough_mode not in ("exact", "accumulator"):
            raise ValueError("hough_mode must be 'exact' or 'accumulator'")
        self.hough_mode = hough_mode
        self.mask = mask
        self.num_cls = num_cls
        self.logger = logger
        self.mask_area = np.sum(self.mask)
        self.bi_cfg = cfg.binary
        self.hough_cfg = cfg.hough_line
        self.dynamic_cfg = cfg.dynamic
        self.stack_maxsize = int(window_sec * fps)
        if not 1 <= self.stack_maxsize <= _lib.MDH_MAX_WINDOW:
            raise ValueError(f"window of {self.stack_maxsize} frames unsupported (1..{_lib.MDH_MAX_WINDOW})")
        if self.bi_cfg.adaptive_bi_thre:
            # the reference reads the sensitivity only in the adaptive case (Detector.py:204-209)
            if self.bi_cfg.sensitivity not in _lib.SENSITIVITY_ID:
                raise KeyError(self.bi_cfg.sensitivity)
            self.std2thre = self.sensitivity_func[self.bi_cfg.sensitivity]
            self.bi_threshold = self.abs_sensitivity[self.bi_cfg.sensitivity]
        else:
            self.bi_threshold = self.bi_cfg.init_value
        self.bi_threshold_float = self.bi_threshold
        self.max_allow_gap = 0.05
        std_roi = select_std_roi(np.asarray(mask), self.bi_cfg.area)
        self.stack = _StackView(self.stack_maxsize, std_roi)
        H, W = mask.shape[:2]
        c = _lib.MdhCfg()
        c.detector = self._native_kind
        c.adaptive_bi_thre = int(bool(self.bi_cfg.adaptive_bi_thre))
        c.init_value = int(self.bi_cfg.init_value)
        c.sensitivity = _lib.SENSITIVITY_ID.get(self.bi_cfg.sensitivity, _lib.SENSITIVITY_ID["normal"])
        c.interval = int(self.bi_cfg.interval)
        for k in range(4):
            c.roi[k] = int(std_roi[k])
        c.hough_threshold = int(self.hough_cfg.threshold)
        c.hough_min_len = int(self.hough_cfg.min_len)
        c.hough_max_gap = int(self.hough_cfg.max_gap)
        c.dy_mask = int(bool(self.dynamic_cfg.dy_mask))
        c.max_batch = int(max_batch)
        c.mask_area = float(self.mask_area)
        c.hough_mode = _lib.MDH_HOUGH_ACCUMULATOR if hough_mode == "accumulator" else _lib.MDH_HOUGH_EXACT
        self._ctx = _lib.Context(H, W, self.stack_maxsize, c, device=device)
        self._H, self._W = H, W
        self._pending = False

    # ---- reference protocol -------------------------------------------------------------------
    def update(self, new_frame: np.ndarray):
        self._ctx.update(new_frame)
        self._pending = True

    def detect(self):
        return [], []

    def visu(self):
        return super().visu()

    # ---- shared plumbing ----------------------------------------------------------------------
    def _absorb_info(self, info):
        self.bi_threshold = int(info["bi_threshold"])
        self.bi_threshold_float = float(info["bi_threshold_float"])
        self.stack.snr = float(info["snr"]) if info["snr"] != 0 else 0
        self.stack.noise_cur_value = float(info["sigma"])
        self.stack.timer = self._ctx.frames_seen
        self.dst_sum = float(info["dst_sum"])
        self.lines_num = int(info["n_raw"])
        self.filtered_line_num = int(info["n_nms"])

    @property
    def dst(self) -> np.ndarray:
        """The binary image of the last processed frame (H,W) uint8 of {0,255}, fetched on demand."""
        return self._ctx.get_binary(-1, 0)

    def close(self):
        self._ctx.close()


class M3Detector(LineDetector):
    """max - mean -> median 3x3 -> threshold -> close 3x3 -> dynamic mask -> HoughLinesP -> NMS
    (Detector.py:302-392), every stage on the GPU except the O(N^2) NMS (N <= 500) in native host code."""

    _native_kind = _lib.MDH_DET_M3

    def __init__(self, window_sec, fps, mask, num_cls, cfg, logger=None, **kw):
        super().__init__(window_sec, fps, mask, num_cls, cfg, logger, **kw)
        self.lines_num = 0
        self.filtered_line_num = 0
        self.dst_sum = 0.0
        self.linesp_ext = np.array([])

    def _package(self, info, raw, nms, prob):
        n_raw, n_nms = int(info["n_raw"]), int(info["n_nms"])
        if n_raw > NUM_LINES_TOOMUCH or n_raw == 0:  # nothing found, or the frame is discarded (Detector.py:358-360)
            return np.array([]), np.array([]), np.zeros((0, self.num_cls))
        cls_pred = np.zeros((n_nms, self.num_cls))
        cls_pred[:, -1] = prob[:n_nms]
        cls_pred[:, 0] = 1 - prob[:n_nms]
        return raw[:n_raw], nms[:n_nms], cls_pred

    def detect(self):
        if not self._pending:
            raise RuntimeError("detect() called without a preceding update()")
        fields, raw, nms, prob = self._ctx.detect_fields()
        self._pending = False
        # (FRAME_INFO_DTYPE's field order; _absorb_info on a record does the same for the batched path)
        n_raw, n_nms, _, self.bi_threshold, self.bi_threshold_float, snr, self.stack.noise_cur_value, self.dst_sum, _ = fields
        self.stack.snr = snr if snr != 0 else 0
        self.stack.timer += 1
        self.lines_num, self.filtered_line_num = n_raw, n_nms
        if n_raw > NUM_LINES_TOOMUCH or n_raw == 0:  # nothing found, or the frame is discarded (Detector.py:358-360)
            self.linesp_ext = _EMPTY
            return _EMPTY, np.zeros((0, self.num_cls))
        cls_pred = np.zeros((n_nms, self.num_cls))
        cls_pred[:, -1] = prob
        cls_pred[:, 0] = 1 - prob
        self.linesp_ext = raw
        return nms, cls_pred

    def detect_batch(self, frames, K: Optional[int] = None, on_device: bool = False):
        """K x (update + detect) in one native call.  frames: (K,H,W) uint8 array, or a device pointer with K.
        Returns (results, info): results[i] = (lines, cls_pred) exactly as K alternating update()/detect() calls
        would have returned; info is the per-frame scalar record array."""
        info, raw, nms, prob = self._ctx.process_batch(frames, K=K, on_device=on_device)
        out = []
        for i in range(len(info)):
            ext, lines, cls_pred = self._package(info[i], raw[i], nms[i], prob[i])
            out.append((lines, cls_pred))
        if len(info):
            self._absorb_info(info[-1])
            self.linesp_ext = ext
        return out, info

    def visu(self):
        """The records of Detector.py:394-448, field for field (tests/golden/g11_visu.json holds what the reference returned).
        `dst` is fetched from the GPU here, on demand: the per-frame loop never pays for it unless the visualiser runs."""
        return visu_records(self.dst, self.stack, self.bi_threshold, self.bi_threshold_float, self.lines_num, self.filtered_line_num,
                            self.dst_sum)


def visu_records(dst, stack, bi_threshold, bi_threshold_float, lines_num, filtered_line_num, dst_sum):
    x1, y1, x2, y2 = [-1] * 4
    if getattr(stack, "std_roi", None):
        x1, y1, x2, y2 = stack.std_roi

    def text(name, body, color="green"):
        return TextVisu(name, position="left-top", color=color, text_list=[TextColorPair(text=body)])

    return [
        ImgVisuAttrs("mix_bg", img=dst // 255, weight=0.5, color="yellow"),
        text("std_value", f"STD:{stack.snr:.4f}"),
        text("bi_value", f"Bi_Threshold: {bi_threshold} (rounded from {bi_threshold_float:.4f})"),
        text("lines_num", f"Line num: {lines_num} (filtered: {filtered_line_num})"),
        text("area_ratio", f"Diff Area: {dst_sum:.2f}%"),
        text("lines_warning", "WARNING: TOO MANY LINES!" if lines_num > 10 else "", color="red"),
        DrawRectVisu("std_roi_area", pair_list=[SquareColorPair(dot_pair=([y1, x1], [y2, x2]))], color="purple"),
    ]


class ClassicDetector(LineDetector):
    """4-frame difference detector (Detector.py:245-299): window hard-wired to 4 frames, all outputs METEOR."""

    classic_max_size = 4
    _native_kind = _lib.MDH_DET_CLASSIC

    def __init__(self, window_sec, fps, mask, num_cls, cfg, logger=None, **kw):
        window_sec = self.classic_max_size / fps
        super().__init__(window_sec, fps, mask, num_cls, cfg, logger, **kw)
        self.linesp_ext = []

    def _package(self, info, raw):
        n_raw = int(info["n_raw"])
        if n_raw > _lib.MDH_MAX_LINES:  # the library keeps MDH_MAX_LINES segments per frame; never hand back a truncated set
            raise RuntimeError("ClassicDetector: more than MDH_MAX_LINES segments in one frame")
        lines = raw[:n_raw] if n_raw else []
        cls_pred = np.zeros((n_raw, self.num_cls))
        cls_pred[:, 0] = 1
        return lines, cls_pred

    def detect(self):
        if not self._pending:
            raise RuntimeError("detect() called without a preceding update()")
        info, raw, _, _ = self._ctx.detect()
        self._pending = False
        self._absorb_info(info)
        self.linesp_ext, cls_pred = self._package(info, raw)
        return self.linesp_ext, cls_pred

    def detect_batch(self, frames, K=None, on_device=False):
        info, raw, _, _ = self._ctx.process_batch(frames, K=K, on_device=on_device)
        out = [self._package(info[i], raw[i]) for i in range(len(info))]
        if len(info):
            self._absorb_info(info[-1])
        return out, info

    def visu(self):
        raise NotImplementedError


available_detectors = [M3Detector, ClassicDetector]


def get_detector(name: str):
    """Class lookup by __name__, as MetLib.get_detector does (MetLib/__init__.py:17-25, 34-46)."""
    table = {cls.__name__: cls for cls in available_detectors}
    if name not in table:
        raise Exception(f"No class named {name} for detector.")
    return table[name]
