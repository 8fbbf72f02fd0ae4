"""Loader-side preprocessing on the GPU (reference: MetLib/imgproc.py Transform, utils.MergeFunction.max,
stacker.max_stacker).  Thin ctypes wrappers over mdh_pre_* / mdh_max_stack; no CPU fallback."""
from __future__ import annotations

import numpy as np

from . import _lib


class Preprocessor:
    """resize(INTER_LINEAR) -> BGR2GRAY -> * mask, then the exposure max-merge (imgproc.py:82-101, utils.py:202-204)."""

    def __init__(self, src_hw, dst_wh=None, mask01: np.ndarray | None = None, device: int = 0):
        self.L = _lib.load()
        self.sh, self.sw = int(src_hw[0]), int(src_hw[1])
        self.dw, self.dh = (self.sw, self.sh) if dst_wh is None else (int(dst_wh[0]), int(dst_wh[1]))
        if mask01 is not None:
            mask01 = np.ascontiguousarray(mask01, dtype=np.uint8)
            if mask01.shape != (self.dh, self.dw):
                raise ValueError(f"mask must be {(self.dh, self.dw)}, got {mask01.shape}")
        self._h = self.L.mdh_pre_create(int(device), self.sh, self.sw, self.dh, self.dw,
                                        mask01.ctypes.data if mask01 is not None else None)
        if not self._h:
            raise _lib.MetDetHipError(f"mdh_pre_create failed: {_lib.last_error()}")

    def __call__(self, bgr_frames: np.ndarray) -> np.ndarray:
        """bgr_frames: (exp_frame, H, W, 3) or (H, W, 3) uint8 -> (dst_h, dst_w) uint8."""
        f = np.ascontiguousarray(bgr_frames, dtype=np.uint8)
        if f.ndim == 3:
            f = f[None]
        if f.shape[1:] != (self.sh, self.sw, 3):
            raise ValueError(f"frames must be (e,{self.sh},{self.sw},3), got {f.shape}")
        out = np.empty((self.dh, self.dw), np.uint8)
        _lib.check(self.L.mdh_pre_run(self._h, f.ctypes.data, f.shape[0], 0, out.ctypes.data, 0), "mdh_pre_run")
        return out

    def run_device(self, src_ptr: int, exp_frame: int, out_ptr: int):
        _lib.check(self.L.mdh_pre_run(self._h, int(src_ptr), int(exp_frame), 1, int(out_ptr), 1), "mdh_pre_run")

    def close(self):
        if getattr(self, "_h", None):
            self.L.mdh_pre_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def max_stack(frames: np.ndarray, device: int = 0) -> np.ndarray:
    """Elementwise max over frames[0..K) (stacker.py:43-49): frames (K, ...) uint8 -> (...) uint8."""
    f = np.ascontiguousarray(frames, dtype=np.uint8)
    out = np.empty(f.shape[1:], np.uint8)
    _lib.check(_lib.load().mdh_max_stack(int(device), f.ctypes.data, f.shape[0], out.nbytes, 0, out.ctypes.data, 0),
               "mdh_max_stack")
    return out
