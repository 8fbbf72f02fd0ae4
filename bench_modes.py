"""Timed entries of bench.py beyond independent streams.

  --mode bands       ONE stream tiled into row bands (BASELINE config 5: 7680x4320 over 8 GPUs).  WORLD_SIZE ranks = one band
                     per GPU (noise sums all-reduced, bit planes through one all-to-all per pass, results gathered: RCCL);
                     one process = `--shards` bands on this device, the exchange done with device-to-device copies.
  --mode timeshard   ONE stream cut into time windows (the sharding north_star names).  WORLD_SIZE ranks = one window per GPU
                     (threshold table all-gathered, results gathered); one process = `--shards` windows one after the other.
  --workload C1e     config-1 substitute end to end: 3840x2160x3 BGR clip in HBM -> loader preprocessing on the GPU
                     (resize 960x540, gray, mask-east) -> detector passes on the device buffers -> native collector.

All three are "strong scaling" entries: the job is one stream of steps x batch frames however many GPUs work on it.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


def _oracle_rows(frames, n, mask, cfg, count):
    """[(thr, lit, n_raw, raw)] of the first `count` frames from the C port of the oracle under the detector preset `cfg`."""
    from metdetpy_amd.utils import select_std_roi
    from oracle import c_oracle as co
    co.lib().om3_set_threads(min(32, len(os.sched_getaffinity(0))))
    # (the preset travels as an argument: `import bench` here is a second copy of the module the script runs as __main__,
    # whose globals -- SENSITIVITY -- were never set)
    orc = co.CM3(n, mask, select_std_roi(mask, cfg.binary.area), sensitivity=cfg.binary.sensitivity, interval=cfg.binary.interval,
                 hough_threshold=cfg.hough_line.threshold, hough_min_len=cfg.hough_line.min_len, hough_max_gap=cfg.hough_line.max_gap)
    rows = []
    for f in frames[:count]:
        orc.update(f)
        raw = orc.detect()
        rows.append((orc.state()[0], orc.lit_count, orc.lines_num, raw))
    return rows


def _parity(rows, info, results, what):
    from oracle import c_oracle as co
    bad = []
    for i, (thr, lit, n_raw, raw) in enumerate(rows):
        ok = int(info[i]["bi_threshold"]) == thr and int(info[i]["lit_count"]) == lit and int(info[i]["n_raw"]) == n_raw
        if ok and 0 < n_raw <= 500:
            nms, prob = co.lineset_nms(raw)
            ok = np.array_equal(np.asarray(results[i][0]).reshape(-1, 4), nms) and np.array_equal(results[i][1][:, -1], prob)
        if not ok:
            bad.append(i)
    return {"frames_checked": len(rows), "mismatching_frames": bad, "verdict": "identical" if not bad else "MISMATCH",
            "checked": f"bi_threshold, lit_count, raw segment count, lineset_nms output of {what} vs oracle/metdet_oracle.c"}


def run_bands(a, D, wl, mask, cfg):
    import bench
    from metdetpy_amd.bands import BandedM3Detector, LocalBands
    H, W, n = wl["H"], wl["W"], wl["n"]
    B = a.batch
    local_shared = False
    frames = bench.make_stream_frames(wl, 0, mask, B, a.unique)  # every rank sees the SAME stream (rank 0's seed)
    uniq = len(frames)
    idx = np.arange(B) % uniq
    if D.world > 1:
        det = BandedM3Detector(1, n, mask, 10, cfg, None, rank=D.rank, world=D.world, group=D.g, device=D.local, max_batch=B)
        plan, ctx = det.plan, det.engine.band
        if os.environ.get("MDH_BENCH_VERBOSE_RANKS") == "1":
            kp = -(-B // D.world)
            print(f"[rank {D.rank}] band rows {plan.bands[D.rank]['own0']}..{plan.bands[D.rank]['own1']} of {H}; "
                  + (f"RCCL communicator {det.comm.describe()}; " if det.comm is not None else "no RCCL communicator (ranks share a device: host buffers through the rendezvous store); ")
                  + f"per pass: all-reduce of 2 x {B // (2 * n) + 1} int64 noise sums (<= {16 * (B // (2 * n) + 1)} B), all-to-all of "
                    f"{kp * plan.band_rows * plan.WW * 8} B per peer ({kp} frames x {plan.band_rows} rows x {plan.WW} words), gather of the results to rank 0",
                  file=sys.stderr, flush=True)
        sub = plan.crop(frames, D.rank)
        ptr = ctx.device_alloc(B * sub[0].size)
        for s in range(0, B, uniq):
            ctx.device_upload(ptr + s * sub[0].size, sub[:min(uniq, B - s)])
        step = lambda: det.detect_batch(ptr, K=B, on_device=True)
        submit = lambda: det.submit(ptr, K=B, on_device=True)
        nb = D.world
    else:
        nb = a.shards
        det = LocalBands(1, n, mask, 10, cfg, nb, device=D.local, max_batch=B)
        plan = det.s.plan
        ptrs = []
        for b, e in enumerate(det.engines):
            sub = plan.crop(frames, b)
            p = e.band.device_alloc(B * sub[0].size)
            for s in range(0, B, uniq):
                e.band.device_upload(p + s * sub[0].size, sub[:min(uniq, B - s)])
            ptrs.append(p)
        step = lambda: det.detect_batch_device(ptrs, B)
        submit = lambda: det.submit_device(ptrs, B)
        local_shared = det.shared
    tally = [0, 0, 0]
    first = None
    for _ in range(max(a.warmup, 1)):
        out = step()
        if first is None:
            first = out
    steps = a.steps if a.steps is not None else 30
    D.fence()
    t0 = time.perf_counter()
    # two passes in flight: the Hough stage of pass p (queued, mdh_hough_planes_submit) runs under the band stages of pass p + 1
    queued = 0
    collect_at = []  # wall time of every collect: a rate that ramps (allocations, clocks) shows here, a uniform one does not
    for s_ in range(steps + 1):
        if s_ < steps:
            submit()
            queued += 1
        if queued == 2 or (s_ == steps and queued):
            res, info = det.collect(arrays=True) if local_shared else det.collect()
            collect_at.append(time.perf_counter() - t0)
            queued -= 1
            if info is not None and (res is not None or local_shared):
                tally[0] += int(info["n_raw"].sum())
                tally[1] += int(info["n_nms"].sum())
                tally[2] += int(info["lit_count"].sum())
    D.fence()
    dt = time.perf_counter() - t0
    dt_all, _ = D.max_time_and_sum(dt, [0, 0, 0])
    extra = {"bands": {"bands": nb, "band_rows": plan.band_rows, "halo_rows": 4, "passes_in_flight": 2,
                       "exchange": "RCCL all-to-all of the bit planes + all-reduce of the noise sums + gather of the results" if D.world > 1
                       else "device-to-device copies on one GPU (stand-in of the all-to-all), every band and every Hough context on this device",
                       "note": "per band and pass: noise share and pixel stages (blocking), the Hough stage of the assembled planes queued "
                               "and collected one pass later; one host thread per band when all bands share one process"}}
    gaps = np.diff([0.0] + collect_at) * 1e3
    extra["bands"]["ms_between_collects"] = {"first": float(gaps[0]), "median": float(np.median(gaps)), "max": float(gaps.max()),
                                             "slowest_at_pass": int(np.argmax(gaps))}
    try:  # which output format the bands' window kernels ended up in (a band that could not allocate its plain buffer stays sparse)
        extra["bands"]["dense_format_bands"] = [bool(e.band.dense_now) for e in det.engines] if D.world == 1 else [bool(ctx.dense_now)]
    except Exception:  # noqa: BLE001
        pass
    if D.rank == 0 and a.parity_frames > 0 and first[0] is not None:
        cnt = min(a.parity_frames, 16, B)
        extra["parity"] = _parity(_oracle_rows(frames[idx[:cnt]], n, mask, cfg, cnt), first[1], first[0], f"the {nb}-band detector")
    det.close()
    return dict(frames=steps * B, dt=dt_all, steps=steps, tot=tally, prof={}, prof_alone={}, frames_sample=None, B=B, uniq=uniq, extra=extra)


def run_timeshard(a, D, wl, mask, cfg):
    """ONE stream cut into time windows (the sharding north_star names).  The stream is resident in HBM (as in the stream
    bench); --host-input reads it from pinned host memory instead (PCIe-inclusive, twice: two sweeps)."""
    import bench
    from metdetpy_amd import _lib
    from metdetpy_amd.timeshard import TimeShard
    H, W, n = wl["H"], wl["W"], wl["n"]
    B = min(a.batch, 1024)
    windows = D.world if D.world > 1 else a.shards
    # resident stream: 16 passes per window by default (a window of a few passes never reaches the pipeline's steady state);
    # host input: 2 (the frames cross PCIe twice)
    steps = a.steps if a.steps is not None else (2 if a.host_input else 16) * windows
    total = steps * B
    base = bench.make_stream_frames(wl, 0, mask, total, a.unique)
    uniq = len(base)
    fb = H * W
    pool = None      # host input: frame i = pool[i % P] in PINNED host memory, so that a pass is a view, not a copy
    dpool = None     # resident input: frame i at dpool + i * fb
    P = min(total, 4 * B)

    def read(lo, hi):
        if pool is None:
            return base[np.arange(lo, hi) % uniq]
        s = lo % P
        if s + (hi - lo) <= P:
            return pool[s:s + (hi - lo)]
        return pool[np.arange(lo, hi) % P]

    ranks = [D.rank] if D.world > 1 else list(range(windows))
    shards = [TimeShard(1, n, mask, 10, cfg, total, read, rank=r, world=windows, device=D.local, max_batch=B,
                        scan_batch=min(B, 256) if a.host_input else B)  # (host frames: every context also holds a ring of its pass)
              for r in ranks]
    import ctypes
    hctx = shards[0].engine.band
    pinned = None
    if a.host_input:
        pinned = hctx.host_alloc(P * fb)
        pool = np.ctypeslib.as_array((ctypes.c_uint8 * (P * fb)).from_address(pinned)).reshape(P, H, W)
        for i in range(P):
            pool[i] = base[i % uniq]
    else:
        # frame i lives at slot i % DP of a pool of DP frames (a whole number of stream periods, so the stream is seamless where
        # the pool wraps) followed by a copy of its first B frames (so that a pass never straddles the end of the pool)
        DP = total if total <= 9 * B else uniq * (-(-8 * B // uniq))
        tail = 0 if DP == total else B
        dpool = hctx.device_alloc((DP + tail) * fb)
        for start in range(0, DP + tail, uniq):  # the stream's period repeated
            cnt = min(uniq, DP + tail - start)
            hctx.device_upload(dpool + start * fb, base[:cnt])
        for s in shards:
            s.read_device = lambda lo, hi: dpool + (lo % DP) * fb
    D.fence()
    t0 = time.perf_counter()
    mine = {}
    if len(shards) > 1:
        # one process, several windows on this device: the first sweep is a chain of short blocking calls per window (launch, wait,
        # 16 bytes per due frame back) -- the windows' chains run side by side from a few host threads (the library releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(min(4, len(shards))) as ex:  # (2 or 8 threads: the same 14-17 ms)
            for got in ex.map(lambda s_: s_.scan(), shards):
                mine.update(got)
    else:
        for s in shards:  # first sweep: the noise sums of the due frames of every window
            mine.update(s.scan())
    t_scan = time.perf_counter() - t0
    if D.world > 1:
        table = {}
        for part in D.all_gather_object(mine):  # the threshold table: 16 bytes of sums per due frame
            table.update(part)
    else:
        table = mine
    part = []
    # second sweep: the detector passes of every window.  Several windows in one process: window r + 1's first passes are queued
    # when window r's last pass has been, BEFORE the host collects r's tail -- the next window's pipeline fills under this one's
    # drain (each window alone never reaches the pipeline's steady state: fill + drain are 3-4 ms of a 16-pass window's 10)
    draining = None
    from metdetpy_amd.timeshard import chains_at
    gens = [s.finish_blocks_steps(table, chain=ch) for s, ch in zip(shards, chains_at(shards, table))]  # (every context positioned before the first pass is queued)
    for s, gen in zip(shards, gens):
        if gen is None:
            part.extend(s.finish(table))
            continue
        for ev in gen:
            if ev == "primed":
                break
        if draining is not None:
            for _ in draining[0]:
                pass
            part.extend(("block",) + b for b in draining[1].blocks)
        for ev in gen:
            if ev == "submitted":
                break
        draining = (gen, s)
    if draining is not None:
        for _ in draining[0]:
            pass
        part.extend(("block",) + b for b in draining[1].blocks)
    gathered = D.gather_object(part)  # the final detection gather (rank 0 gets the list, the others None)
    D.fence()
    dt = time.perf_counter() - t0
    dt_all, _ = D.max_time_and_sum(dt, [0, 0, 0])
    first = read(0, min(64, total)).copy()
    if pinned is not None:
        del pool
        pool = None
        hctx.host_free(pinned)
    if dpool is not None:
        hctx.device_free(dpool)
    for s in shards:
        s.close()
    tally = [0, 0, 0]
    extra = {"timeshard": {"windows": windows, "frames": total, "warm_up_frames_per_window": 2 * (n - 1), "first_sweep_s": t_scan,
                           "input": "pinned host memory (PCIe-inclusive, twice: two sweeps)" if a.host_input else "resident in HBM",
                           "note": "first sweep = noise sums of the due frames of every window (all-gathered table, 16 B per due frame); second "
                                   "sweep = the pipelined detector passes of every window on a context positioned with mdh_seek "
                                   "(+ mdh_inject_sums for the sigmas that fall into the refilling window, Hough muted over the warm-up)"}}
    if D.rank == 0:
        info = np.zeros(total, _lib.FRAME_INFO_DTYPE)
        results = [None] * total
        from metdetpy_amd.timeshard import unpack_block
        empty = (np.array([]), np.zeros((0, 10)))
        for chunk in gathered:
            for rec in chunk:
                if rec[0] == "block":
                    f0, binfo = rec[1], rec[2]
                    info[f0:f0 + len(binfo)] = binfo
                    if f0 < 64:  # per-frame (lines, cls_pred) only where the parity leg looks
                        for i, _fi, lines, pred in unpack_block(rec[1:], 10):
                            results[i] = (lines, pred)
                else:
                    i, fi, lines, pred = rec
                    info[i] = fi
                    results[i] = (lines, pred)
        results = [r if r is not None else empty for r in results]
        tally = [int(info["n_raw"].sum()), int(info["n_nms"].sum()), int(info["lit_count"].sum())]
        if a.parity_frames > 0:
            cnt = min(a.parity_frames, 64, total)
            extra["parity"] = _parity(_oracle_rows(first, n, mask, cfg, cnt), info, results, "the time-sharded detector")
    return dict(frames=total, dt=dt_all, steps=steps, tot=tally, prof={}, prof_alone={}, frames_sample=None, B=B, uniq=uniq, extra=extra)


def run_c1e(a, D, wl, mask, cfg):
    """3840x2160x3 BGR frames resident in HBM -> mdh_pre_run_batch (resize to 960x540 + BGR2GRAY + mask) -> mdh_submit_batch on the
    device buffer -> mdh_collect_batch -> native collector, several detector passes in flight behind the preprocessing."""
    import bench
    from metdetpy_amd import M3Detector, _lib
    from metdetpy_amd.collector import MeteorCollector
    from metdetpy_amd.imgproc import Preprocessor
    from metdetpy_amd.metstruct import CollectorCfg, MeteorCfg, RecheckSwitch, RuntimeParams
    from metdetpy_amd.synth import SkyStream
    H, W, n = wl["H"], wl["W"], wl["n"]
    SH, SW = wl["src"]
    B = a.batch
    uniq = min(B, 50 if a.unique >= 50 else max(a.unique, 2 * n))
    # the clip: the 960x540 sky stream blown up 4x with a fixed 4K dither pattern on top, three channels with different gains
    rng = np.random.default_rng(wl["seed"] + 7)
    dither = rng.integers(-3, 4, (SH, SW, 3), dtype=np.int16)
    gain = np.array([0.92, 1.0, 1.06], np.float32)
    small = SkyStream(H, W, wl["seed"] + D.rank).take(uniq)
    det = M3Detector(1, n, mask, 10, cfg, None, device=D.local, max_batch=B)
    ctx = det._ctx
    pre = Preprocessor((SH, SW), (W, H), mask, device=D.local)
    fbytes = SH * SW * 3
    src = ctx.device_alloc(B * fbytes)
    clip = np.empty((uniq, SH, SW, 3), np.uint8)
    for k in range(uniq):
        up = np.repeat(np.repeat(small[k], SH // H, 0), SW // W, 1).astype(np.float32)
        clip[k] = np.clip(np.rint(up[..., None] * gain) + dither, 0, 255).astype(np.uint8)
    for s in range(0, B, uniq):
        ctx.device_upload(src + s * fbytes, clip[:min(uniq, B - s)])
    nbuf = ctx.max_in_flight + 1
    outs = [ctx.device_alloc(B * H * W) for _ in range(nbuf)]
    ccfg = CollectorCfg(MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})
    rt = RuntimeParams(fps=float(n), exp_frame=1, eq_fps=float(n), eq_int_fps=n, exp_time=1.0 / n, runtime_size=[W, H],
                       raw_size=[SW, SH], positive_category_list=["METEOR", "RED_SPRITE"])
    col = MeteorCollector(ccfg, rt)
    info = np.zeros(B, _lib.FRAME_INFO_DTYPE)
    nms = np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32)
    prob = np.zeros((B, _lib.MDH_MAX_LINES), np.float64)
    tally = [0, 0, 0]
    state = dict(sub=0, col=0, pre_s=0.0)
    first = {}

    def collect():
        ctx.collect_into(info, None, nms, prob)
        col.update_batch(state["col"] * B, state["col"] * B, info, nms, prob)
        if not first:
            first["info"], first["nms"], first["prob"] = info.copy(), nms.copy(), prob.copy()
        state["col"] += 1
        tally[0] += int(info["n_raw"].sum())
        tally[1] += int(info["n_nms"].sum())
        tally[2] += int(info["lit_count"].sum())

    def run(nsteps):
        for _ in range(nsteps):
            if ctx.in_flight == ctx.max_in_flight:
                collect()
            buf = outs[state["sub"] % nbuf]
            t = time.perf_counter()
            pre.batch_device(src, B, 1, buf)  # blocking: returns when the pass's frames are preprocessed
            state["pre_s"] += time.perf_counter() - t
            ctx.submit(buf, B, True)
            state["sub"] += 1
        while ctx.in_flight:
            collect()

    run(max(a.warmup, 1))
    steps = a.steps if a.steps is not None else 400
    tally[:] = [0, 0, 0]
    state["pre_s"] = 0.0
    D.fence()
    t0 = time.perf_counter()
    run(steps)
    D.fence()
    dt = time.perf_counter() - t0
    dt_all, tot = D.max_time_and_sum(dt, tally)
    pre_ms = state["pre_s"] / steps * 1e3
    # Priced on the bytes the kernel MUST move: the source rows the resize's vertical taps touch (2 of every 4 at this 4x
    # reduction -- whole rows: the horizontal taps use bytes of every cache line of a touched row) + mask + output.  Round 3 priced
    # it on every source byte once (`frac_every_source_byte`, kept for comparison): half of those the kernel never reads.
    must = B * pre.traffic_model(1)
    live = None
    if D.rank == 0 and D.world == 1 and not a.no_pmc:
        live = bench.pmc_traffic_live(a, False, kern="k_preprocess")
    extra = {"roofline": {"bound": "hbm", "kernel": "preprocess (resize INTER_LINEAR 4x down + BGR2GRAY + mask, one launch per pass, source rows staged through LDS)",
                          "peak": bench.HBM_PEAK_GBS, "unit": "GB/s", "achieved": must / (pre_ms * 1e-3) / 1e9,
                          "frac": must / (pre_ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS, "bytes_per_launch": must, "launch_ms": pre_ms, "launches": steps,
                          "bytes_model": f"{B} frames x (source rows touched x row bytes + mask + output) = {B} x {pre.traffic_model(1)}",
                          "frac_every_source_byte": B * fbytes / (pre_ms * 1e-3) / 1e9 / bench.HBM_PEAK_GBS,
                          "traffic": live[0] if live else None, "traffic_source": live[1] if live else None,
                          "traffic_over_compulsory": (live[0] / must) if live and live[0] else None,
                          "launch_ms_note": "host-timed blocking call (launch + kernel + stream sync)"},
             "records": len(col.met_exporter.meteor_list)}
    if D.rank == 0 and a.parity_frames > 0:
        from oracle import m3_oracle as mo
        cnt = min(a.parity_frames, 24, uniq)
        gray = np.stack([mo.preprocess_frame(clip[k], (W, H), mask) for k in range(cnt)])
        extra["parity"] = _parity(_oracle_rows(gray, n, mask, cfg, cnt), first["info"],
                                  [(first["nms"][i][:int(first["info"][i]["n_nms"])],
                                    np.concatenate([np.zeros((int(first["info"][i]["n_nms"]), 9)), first["prob"][i][:int(first["info"][i]["n_nms"]), None]], 1))
                                   for i in range(cnt)], "BGR clip -> oracle loader chain (resize, gray, mask) -> oracle detector")
    col.clear()
    for p in outs + [src]:
        ctx.device_free(p)
    pre.close()
    det.close()
    return dict(frames=steps * B * D.world, dt=dt_all, steps=steps, tot=tot, prof={}, prof_alone={}, frames_sample=None, B=B, uniq=uniq, extra=extra)
