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

    def batch(self, bgr_frames: np.ndarray, exp_frame: int = 1) -> np.ndarray:
        """bgr_frames: (n_out * exp_frame, H, W, 3) uint8 -> (n_out, dst_h, dst_w) uint8, one launch."""
        f = np.ascontiguousarray(bgr_frames, dtype=np.uint8)
        if f.ndim != 4 or f.shape[1:] != (self.sh, self.sw, 3) or len(f) % exp_frame:
            raise ValueError(f"frames must be (n*{exp_frame},{self.sh},{self.sw},3), got {f.shape}")
        n_out = len(f) // exp_frame
        out = np.empty((n_out, self.dh, self.dw), np.uint8)
        _lib.check(self.L.mdh_pre_run_batch(self._h, f.ctypes.data, n_out, int(exp_frame), 0, out.ctypes.data, 0), "mdh_pre_run_batch")
        return out

    def batch_device(self, src_ptr: int, n_out: int, exp_frame: int, out_ptr: int):
        """Device to device: out_ptr (n_out * dst_h * dst_w bytes) can be handed to the detector with on_device=True."""
        _lib.check(self.L.mdh_pre_run_batch(self._h, int(src_ptr), int(n_out), int(exp_frame), 1, int(out_ptr), 1), "mdh_pre_run_batch")

    def traffic_model(self, exp_frame: int = 1) -> int:
        """Bytes a launch must move per output frame: the source rows the resize touches + mask + output (mdh_pre_traffic_model)."""
        return int(self.L.mdh_pre_traffic_model(self._h, int(exp_frame)))

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


def load_mask(image: np.ndarray | None, kind: str = "jpg", dsize_wh=None, grayscale: bool = True, device: int = 0) -> np.ndarray:
    """fileio.load_mask (fileio.py:250-292) on a decoded mask image (decoding is the caller's: the reference uses cv2.imread).
    kind "jpg": BGR image -> resize -> BGR2GRAY -> (> 128 -> 1);  kind "png": BGRA image -> alpha channel -> resize ->
    (> 128 -> 0, else 1).  image None: all ones of size dsize_wh.  The resize / gray arithmetic runs in the same GPU kernel
    as the frame preprocessing (a gray of three equal channels is the channel itself: the 15-bit weights sum to 2^15)."""
    if image is None:
        if dsize_wh is None:
            raise ValueError("opencv_resize is required when mask_fname is empty!")
        m = np.ones((int(dsize_wh[1]), int(dsize_wh[0])), np.uint8)
    else:
        img = np.ascontiguousarray(image, dtype=np.uint8)
        if kind == "jpg":
            if img.ndim != 3 or img.shape[2] != 3:
                raise ValueError("a .jpg mask must be a BGR image")
            src = img
        elif kind == "png":
            if img.ndim != 3:
                raise ValueError("a .png mask must have an alpha channel")
            src = np.repeat(img[:, :, -1:], 3, axis=2)
        else:
            raise ValueError("mask kind must be 'jpg' or 'png'")
        pre = Preprocessor(src.shape[:2], dsize_wh, None, device=device)
        g = pre(src)
        pre.close()
        m = (g > 128).astype(np.uint8) if kind == "jpg" else (g <= 128).astype(np.uint8)
    return m if grayscale else np.repeat(m[:, :, None], 3, axis=2)
