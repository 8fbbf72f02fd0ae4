import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metdetpy_amd import M3Detector
from tests.helpers import binary_cfg, frames_normal
shape, n = (72, 256), 4
mask = np.ones(shape, np.uint8)
frames = np.stack(frames_normal(1, 40, shape, 30, 2.0) + frames_normal(2, 56, shape, 30, 9.0) + frames_normal(3, 40, shape, 30, 2.0))
det = M3Detector(1, n, mask, 10, binary_cfg("normal"), None, max_batch=8)
for t0 in range(0, len(frames), 8):
    res, info = det.detect_batch(frames[t0:t0 + 8])
    print(t0, "thr", info["bi_threshold"].tolist(), "lit", info["lit_count"].tolist(), "dens", det._ctx.last_density, "dense_passes", det._ctx.dense_passes)
