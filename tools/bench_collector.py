#!/usr/bin/env python3
"""Updates per second of the Python MeteorCollector and the native one (host only, no GPU) on detector-shaped results:
one streak of ~12 frames every 50 frames, 2 segments per frame (the bench stream's shape)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from metdetpy_amd import _lib
from metdetpy_amd.collector import NativeMeteorCollector
from oracle.collector_oracle import MeteorCollector
from metdetpy_amd.metstruct import CollectorCfg, MeteorCfg, RecheckSwitch, RuntimeParams


def main():
    total, K = 40960, 2048
    rt = RuntimeParams(25.0, 1, 25.0, 25, 0.04, [1920, 1080], [1920, 1080], ["METEOR"])
    cfg = CollectorCfg(MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})
    rng = np.random.default_rng(1)
    info = np.zeros(total, _lib.FRAME_INFO_DTYPE)
    nms = np.zeros((total, _lib.MDH_MAX_LINES, 4), np.int32)
    prob = np.zeros((total, _lib.MDH_MAX_LINES), np.float64)
    for t0 in range(10, total - 20, 50):
        p, ang = rng.uniform(300, 1500, 2), rng.uniform(0, 2 * np.pi)
        v = 24 * np.array([np.cos(ang), np.sin(ang)])
        for k in range(12):
            f = t0 + k
            info[f]["n_raw"], info[f]["n_nms"] = 3, 2
            for j in range(2):
                a = p + v * k + rng.normal(0, 1.5, 2)
                nms[f, j] = np.rint([a[0], a[1], a[0] + v[0] * 2, a[1] + v[1] * 2])
                prob[f, j] = rng.uniform(0, 0.2)
    fed = int(np.count_nonzero(info["n_nms"]) + np.count_nonzero((info["n_nms"] == 0) & (np.arange(total) % 25 == 0)))
    out = {"frames": total, "collector_updates": fed}
    a = MeteorCollector(cfg, rt, None, None)
    t0 = time.perf_counter()
    for f in range(total):
        n = int(info[f]["n_nms"])
        if n or f % rt.eq_int_fps == 0:
            cates = np.zeros((n, 10))
            cates[:, -1] = prob[f, :n]
            cates[:, 0] = 1 - prob[f, :n]
            a.update(f, nms[f, :n], cates)
    a.clear()
    dt = time.perf_counter() - t0
    out["python"] = {"seconds": dt, "frames_per_s": total / dt, "updates_per_s": fed / dt, "records": len(a.met_exporter.meteor_list)}
    b = NativeMeteorCollector(cfg, rt, None, None)
    t0 = time.perf_counter()
    for p0 in range(0, total, K):
        b.update_batch(p0, p0, info[p0:p0 + K], nms[p0:p0 + K], prob[p0:p0 + K])
    b.clear()
    dt = time.perf_counter() - t0
    out["native_batch"] = {"seconds": dt, "frames_per_s": total / dt, "updates_per_s": fed / dt,
                           "records": len(b.met_exporter.meteor_list)}
    out["identical_records"] = [r.to_dict() for r in a.met_exporter.meteor_list] == [r.to_dict() for r in b.met_exporter.meteor_list]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
