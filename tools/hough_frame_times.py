"""Per-frame Hough cost of one bench pass (MDH_HOUGH_DEBUG=1): points, tier, set-up / PPHT time, segments."""
import os, sys, ctypes as C
os.environ["MDH_HOUGH_DEBUG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.synth import SkyStream

WL = os.environ.get("WL", "C2")
H, W, n, B, uniq = {"C2": (1080, 1920, 8, 1024, 250), "C3": (2160, 3840, 16, 256, 100), "C5": (4320, 7680, 8, 64, 50), "C1s": (540, 960, 8, 2048, 250)}[WL]
n = int(os.environ.get("WINDOW", n))  # WINDOW=25: the reference's default window length on the same stream
mask = np.ones((H, W), np.uint8)
bench.SENSITIVITY = bench.WORKLOADS[WL].get("sensitivity", "normal")
STREAM = os.environ.get("STREAM", "sky")  # bench.py --stream: sky | noisy | hot | cloud | sparse | dense
frames = SkyStream(H, W, bench.WORKLOADS[WL]["seed"], **bench.STREAMS[STREAM]["kw"]).take(uniq)
det = M3Detector(1, n, mask, 10, bench.make_cfg(), None, max_batch=B)
ctx = det._ctx
pool = ctx.device_alloc(B * H * W)
for start in range(0, B, uniq):
    ctx.device_upload(pool + start * H * W, frames[:min(uniq, B - start)])
info = np.zeros(B, _lib.FRAME_INFO_DTYPE)
for _ in range(3):
    ctx.process_batch_into(pool, B, True, info, None, None, None)
dbg2 = np.zeros((2, B, 8), np.int64)
rc = ctx.L.mdh_debug_hough(C.c_void_p(ctx.handle), dbg2.ctypes.data_as(C.POINTER(C.c_longlong)), B)
assert rc == 0
dbg, ph = dbg2[0], dbg2[1]
glob_alive, glob_all = (dbg[:, 1] >> 8) & 0xFF, (dbg[:, 1] >> 16) & 0xFF  # hybrid tier: angles on the pooled global accumulator
dbg[:, 1] &= 0xFF
act = dbg[:, 0] > 0
t = dbg[:, 3] / 100.0  # us
print("frames with points", int(act.sum()), "of", B, " tiers:", {int(k): int((dbg[act, 1] == k).sum()) for k in np.unique(dbg[act, 1])})
print("points: median %d  p90 %d  max %d" % tuple(np.percentile(dbg[act, 0], [50, 90, 100])))
print("PPHT us: median %.1f  p90 %.1f  p99 %.1f  max %.1f   sum %.0f" % (*np.percentile(t[act], [50, 90, 99, 100]), t[act].sum()))
print("set-up us: median %.1f max %.1f" % tuple(np.percentile(dbg[act, 2] / 100.0, [50, 100])))
span = (dbg[act, 6].max() - dbg[act, 7].min()) / 100.0
print("first start -> last end: %.1f us" % span)
order = np.argsort(-t)[:12]
for i in order:
    print("frame %4d  N %5d  tier %d  ppht %.1f us  segs %d  cells %d  start +%.1f us" % (i, dbg[i, 0], dbg[i, 1], t[i], dbg[i, 4], dbg[i, 5], (dbg[i, 7] - dbg[act, 7].min()) / 100.0))
# time vs N
for lo, hi in ((1, 50), (50, 200), (200, 500), (500, 1000), (1000, 4000), (4000, 100000)):
    m = act & (dbg[:, 0] >= lo) & (dbg[:, 0] < hi)
    if m.any():
        print("N in [%d,%d): %d frames, mean ppht %.1f us, mean segs %.1f" % (lo, hi, m.sum(), t[m].mean(), dbg[m, 4].mean()))

# phase totals of the three-wave kernel (shader cycles, thread 0's view), summed over the frames with points
names = ["steps (table, list, mask)", "votes", "hits + barrier", "trigger (take back, arg-max)", "line walk", "line clear", "un-vote + emit", "lines"]
cnt = ph[:, 7].copy()
ph[:, 7] = cnt & 0xFFFF
votes, subs = (cnt >> 16) & 0xFFFFFF, cnt >> 40
tot = ph[act].sum(axis=0)
cyc = tot[:7].sum()
print("per frame: votes %.1f  sub-groups %.1f  steps of 64 points %.1f" % (votes[act].mean(), subs[act].mean(), np.ceil(dbg[act, 0] / 64).mean()))
if cyc > 0:
    print("phase share of PPHT cycles:", {n: round(float(v) / cyc, 3) for n, v in zip(names[:7], tot[:7])}, " line events per frame: %.1f" % (tot[7] / act.sum()),
          " cycles per frame: %.0f" % (cyc / act.sum()), " cycles per line event (walk+clear+un-vote): %.0f" % (tot[4:7].sum() / max(tot[7], 1)))
# the same split per tier (0 small LDS, 1 large LDS, 2 hybrid)
print("hybrid frames: angles outside LDS  mean %.1f  of them still in the replay (could reach the threshold)  mean %.1f  max %d" % (
    (glob_all[act & (dbg[:, 1] == 2)].mean() if (act & (dbg[:, 1] == 2)).any() else 0), (glob_alive[act & (dbg[:, 1] == 2)].mean() if (act & (dbg[:, 1] == 2)).any() else 0),
    glob_alive[act].max() if act.any() else 0))
for tier in np.unique(dbg[act, 1]):
    m = act & (dbg[:, 1] == tier)
    tt = ph[m].sum(axis=0)
    c = tt[:7].sum()
    if c > 0:
        print("tier %d: %d frames, mean N %.0f, mean ppht %.1f us, set-up %.1f us, votes %.0f, sub-groups %.1f, line events %.1f, cells %.0f | share:" % (
            tier, m.sum(), dbg[m, 0].mean(), t[m].mean(), (dbg[m, 2] / 100.0).mean(), votes[m].mean(), subs[m].mean(), tt[7] / m.sum(), dbg[m, 5].mean()),
            {n.split(" ")[0]: round(float(v) / c, 3) for n, v in zip(names[:7], tt[:7])}, " cycles per frame %.0f" % (c / m.sum()))
