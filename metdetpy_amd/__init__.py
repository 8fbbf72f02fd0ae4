"""metdetpy_amd -- MI355X-native implementation of MetDetPy's per-frame line-detector hot path.

Python host code (this package) calls hand-written HIP kernels for gfx950 through the C-ABI declared in
include/metdet_hip.h (ctypes; no torch on the data path).  The plugin surface mirrors the reference:
`get_detector("M3Detector")(window_sec, fps, mask, num_cls, cfg, logger)`, `.update(frame)`, `.detect()`.
"""
from .detector import (BaseDetector, ClassicDetector, LineDetector, M3Detector, available_detectors, get_detector)
from .metstruct import (BinaryCfg, BinaryCoreCfg, CollectorCfg, DynamicCfg, HoughLineCfg, MeteorCfg, RuntimeParams)
from .utils import get_num_class

__all__ = [
    "BaseDetector", "LineDetector", "M3Detector", "ClassicDetector", "available_detectors", "get_detector", "BinaryCfg",
    "BinaryCoreCfg", "HoughLineCfg", "DynamicCfg", "CollectorCfg", "MeteorCfg", "RuntimeParams", "get_num_class"
]
