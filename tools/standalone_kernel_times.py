"""Stand-alone cost of every stage: one pass in flight at a time (nothing overlaps), per-kernel HIP-event spans."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.synth import SkyStream

H, W, n, B, uniq = 1080, 1920, int(os.environ.get("MDH_PROBE_N", "8")), 1024, 256  # MDH_PROBE_N: window length
mask = np.ones((H, W), np.uint8)
frames = SkyStream(H, W, bench.WORKLOADS["C2"]["seed"]).take(uniq)
det = M3Detector(1, n, mask, 10, bench.make_cfg(), None, max_batch=B)
ctx = det._ctx
pool = ctx.device_alloc(B * H * W)
for start in range(0, B, uniq):
    ctx.device_upload(pool + start * H * W, frames)
info = np.zeros(B, _lib.FRAME_INFO_DTYPE)
for _ in range(3):
    ctx.process_batch_into(pool, B, True, info, None, None, None)
ctx.profile_enable(True)
ctx.profile_read(True)
t0 = time.perf_counter()
S = 20
for _ in range(S):
    ctx.process_batch_into(pool, B, True, info, None, None, None)
dt = (time.perf_counter() - t0) / S
prof = ctx.profile_read(True)
print("sync pass: %.3f ms" % (dt * 1e3))
print({k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()})
