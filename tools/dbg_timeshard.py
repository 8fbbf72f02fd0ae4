import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, bench_modes
from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.timeshard import TimeShard
wl = bench.WORKLOADS["C2"]
H, W, n = wl["H"], wl["W"], wl["n"]
mask = np.ones((H, W), np.uint8)
cfg = bench.make_cfg()
B = int(os.environ.get("B", "1024"))
total = int(os.environ.get("TOTAL", "4096"))
windows = int(os.environ.get("WINDOWS", "1"))
base = bench.make_stream_frames(wl, 0, mask, total, 250)
uniq = len(base)
fb = H * W
rows = bench_modes._oracle_rows(base[np.arange(64) % uniq], n, mask, "normal", 64)

def report(tag, info):
    bad = [i for i, (thr, lit, n_raw, raw) in enumerate(rows)
           if not (int(info[i]["bi_threshold"]) == thr and int(info[i]["lit_count"]) == lit and int(info[i]["n_raw"]) == n_raw)]
    print(tag, "bad frames:", bad[:12])
    for i in bad[:3]:
        print("   frame", i, "gpu thr/lit/nraw", int(info[i]["bi_threshold"]), int(info[i]["lit_count"]), int(info[i]["n_raw"]), " oracle", rows[i][:3])

# (b) plain detector, device-resident frames, one big pass and small passes
det = M3Detector(1, n, mask, 10, cfg, None, max_batch=B)
ctx = det._ctx
dpool = ctx.device_alloc(total * fb)
for start in range(0, total, uniq):
    cnt = min(uniq, total - start)
    ctx.device_upload(dpool + start * fb, base[:cnt])
info, raw, nms, prob = ctx.process_batch(dpool, K=B, on_device=True)
report(f"plain on_device K={B}", info)
ctx.seek(0, thr=5, thr_float=5.0)
info, raw, nms, prob = ctx.process_batch(dpool, K=B, on_device=True)
report(f"plain on_device after seek K={B}", info)
ctx.seek(0, thr=5, thr_float=5.0)
info, raw, nms, prob = ctx.process_batch(base[np.arange(B) % uniq], K=B)
report(f"plain host after seek K={B}", info)
det.close()
# (a) TimeShard
for resident in (True, False):
    read = lambda lo, hi: base[np.arange(lo, hi) % uniq]
    shards = [TimeShard(1, n, mask, 10, cfg, total, read, rank=r, world=windows, max_batch=B, scan_batch=min(B, 256)) for r in range(windows)]
    if resident:
        for s in shards:
            s.read_device = lambda lo, hi: dpool + lo * fb
    table = {}
    for s in shards:
        table.update(s.scan())
    out = []
    for s in shards:
        out += s.finish(table)
    info = np.zeros(total, _lib.FRAME_INFO_DTYPE)
    for i, fi, lines, pred in out:
        info[i] = fi
    report(f"timeshard resident={resident}", info)
    for s in shards:
        s.close()
