import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from metdetpy_amd import M3Detector
from metdetpy_amd.synth import SkyStream
from oracle import c_oracle as co, m3_oracle as mo
from tests.helpers import binary_cfg, frames_normal, oracle_kwargs
shape, n = (72, 256), 4
mask = np.ones(shape, np.uint8)
cfg = binary_cfg("normal")
s = SkyStream(*shape, 3, speed=5, streak_period=9, streak_life=6)
frames = np.stack(frames_normal(1, 40, shape, 30, 2.0) + frames_normal(2, 56, shape, 30, 9.0) + [s.next() for _ in range(80)])
det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
bad = 0
for t0 in range(0, len(frames), 8):
    before = det._ctx.dense_passes
    res, info = det.detect_batch(frames[t0:t0 + 8])
    mode = det._ctx.dense_passes - before
    for i in range(len(info)):
        orc.update(frames[t0 + i]); orc.detect()
        if int(info[i]["lit_count"]) != orc.lit_count or int(info[i]["bi_threshold"]) != orc.bi_threshold:
            bad += 1
            if bad < 6: print("MISMATCH frame", t0 + i, "dense" if mode else "sparse", int(info[i]["lit_count"]), orc.lit_count, int(info[i]["bi_threshold"]), orc.bi_threshold)
    if t0 + 8 > 97 >= t0 or bad:
        d = det.dst; 
        if not np.array_equal(d, orc.dst):
            yy, xx = np.nonzero(d != orc.dst); print(" last frame of pass", t0, "dst differs at", list(zip(yy[:8].tolist(), xx[:8].tolist())), "gpu", d[yy[:8], xx[:8]].tolist())
        a = det._ctx.get_binary(-1, 1)
        if not np.array_equal(a, orc.act):
            yy, xx = np.nonzero(a != orc.act); print(" act differs at", list(zip(yy[:8].tolist(), xx[:8].tolist())), "gpu", a[yy[:8], xx[:8]].tolist())
print("env MDH_DENSE", os.environ.get("MDH_DENSE"), "mismatching frames", bad, "dense passes", det._ctx.dense_passes)
