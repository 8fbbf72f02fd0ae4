import os, sys, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metdetpy_amd import _lib
from tests.band_oracle_engine import pack_rows
from tests.helpers import binary_cfg
from tests.test_gpu_hough_accumulator import hough_context, line_image
H, W = 240, 320
cfg = binary_cfg()
cases = [(90.0, 119.63, 230.0, 120.37), (90.0, 120.0, 230.0, 120.0), (100.0, 50.0, 200.0, 150.0), (160.0, 40.0, 160.0, 200.0), (230.0, 120.0, 90.0, 120.0)]
imgs = [line_image(H, W, *c) for c in cases]
K = len(imgs)
planes = np.stack([pack_rows(im, (W + 63) // 64) for im in imgs])[None]
for mode in (0, 1):
    ctx = hough_context(H, W, K, mode, cfg, H * W)
    info, raw, nms, prob = ctx.hough_planes(planes, K, 1, H)
    ctx.close()
    for j in range(K):
        n = int(info[j]["n_raw"])
        print("mode", mode, "case", j, "lit", int(info[j]["lit_count"]), "gap", float(info[j]["hough_gap"]), "n_raw", n, raw[j, :min(n, 6)].tolist())
