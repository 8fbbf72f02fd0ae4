"""Summary of a rocprofv3 --kernel-trace CSV: busy share of the GPU, per-kernel totals, per-queue busy time and the largest idle gaps,
over the steady-state part of the run (default: the last 40 % of the timeline).   python tools/trace_summary.py r_kernel_trace.csv [from] [to]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = list(csv.DictReader(open(path)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:44], r["Queue_Id"]) for r in rows))
t0, t1 = ev[0][0], max(e[1] for e in ev)
a, b = t0 + (t1 - t0) * lo, t0 + (t1 - t0) * hi
sel = [e for e in ev if e[0] >= a and e[1] <= b]
span = (sel[-1][1] - sel[0][0]) / 1e6
busy, cur_s, cur_e = 0, None, None
gaps = []
for s, e, n, q in sel:
    if cur_e is None:
        cur_s, cur_e, last = s, e, n
    elif s <= cur_e:
        if e > cur_e:
            cur_e, last = e, n
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, last, n, (cur_e - sel[0][0]) / 1e6))
        cur_s, cur_e, last = s, e, n
busy += cur_e - cur_s
print(f"window {span:.2f} ms, {len(sel)} kernels, GPU busy (union of kernel spans) {busy / 1e6:.2f} ms = {busy / 1e6 / span:.1%}")
tot, cnt = defaultdict(int), defaultdict(int)
qb = defaultdict(int)
for s, e, n, q in sel:
    tot[n] += e - s
    cnt[n] += 1
    qb[q] += e - s
for n in sorted(tot, key=lambda k: -tot[k])[:16]:
    print(f"  {n:44s} {cnt[n]:6d} x {tot[n] / cnt[n] / 1e3:9.1f} us = {tot[n] / 1e6:8.2f} ms ({tot[n] / 1e6 / span:.1%} of the window)")
print("  per queue busy:", {q: f"{v / 1e6 / span:.0%}" for q, v in sorted(qb.items())})
gaps.sort(reverse=True)
print("  idle gaps: total %.2f ms; largest:" % (sum(g[0] for g in gaps) / 1e6))
for g in gaps[:8]:
    print(f"    {g[0] / 1e3:8.1f} us at +{g[3]:.2f} ms  after {g[1]}  before {g[2]}")
