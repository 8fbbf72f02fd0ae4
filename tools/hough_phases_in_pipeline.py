"""PPHT phase cycles of the last pass of a PIPELINED run (window kernels of later passes running next to it) against a run of
isolated passes: which phases stretch when the memory system is busy?"""
import os, sys, ctypes as C
os.environ["MDH_HOUGH_DEBUG"] = "1"
MID = os.environ.get("MDH_HOUGH_DEBUG_PASS")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.synth import SkyStream

H, W, n, B, uniq = 1080, 1920, 8, 1024, 256
frames = SkyStream(H, W, bench.WORKLOADS["C2"]["seed"]).take(uniq)
det = M3Detector(1, n, np.ones((H, W), np.uint8), 10, bench.make_cfg(), None, max_batch=B)
ctx = det._ctx
pool = ctx.device_alloc(B * H * W)
for start in range(0, B, uniq):
    ctx.device_upload(pool + start * H * W, frames)
info = np.zeros(B, _lib.FRAME_INFO_DTYPE)
names = ["draws", "list", "mask", "votes", "keys", "trigger", "line"]


def phases(tag):
    dbg = np.zeros((2, B, 8), np.int64)
    assert ctx.L.mdh_debug_hough(C.c_void_p(ctx.handle), dbg.ctypes.data_as(C.POINTER(C.c_longlong)), B) == 0
    act = dbg[0][:, 0] > 0
    ph = dbg[1][act].sum(axis=0)
    groups = ph[7]
    print(tag, "frames", int(act.sum()), "cycles/group", round(ph[:7].sum() / groups), {k: round(v / groups) for k, v in zip(names, ph[:7])},
          "PPHT us median", float(np.median(dbg[0][act, 3] / 100.0)))


if MID is None:
    for _ in range(3):
        ctx.process_batch_into(pool, B, True, info, None, None, None)
    phases("isolated :")
for rep in range(1 if MID else 2):
    for _ in range(12):
        if ctx.in_flight == ctx.max_in_flight:
            ctx.collect_into(info, None, None, None)
        ctx.submit(pool, B, True)
    # leave the youngest passes in flight while the one whose stamps we read finishes: collect all but keep order
    while ctx.in_flight:
        ctx.collect_into(info, None, None, None)
    phases("pipelined:")
