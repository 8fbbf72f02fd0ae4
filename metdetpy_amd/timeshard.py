"""One stream cut into contiguous time windows across GPUs (SURVEY 8e "time windows of one stream"; the sharding
BASELINE.json's north_star names: embarrassingly parallel, a collective only for the threshold table and the final gather).

Rank r owns frames [s_r, e_r).  Two things tie a frame to its past:

  * the n-frame max / mean window and the n-deep ring of binaries of the dynamic mask (Detector.py:225-242, 327-340):
    rank r starts 2(n-1) frames early and drops the results of those warm-up frames -- after them every piece of pixel
    state is exactly what a single detector would hold;
  * the binary threshold: T(t) = round(a * snr(t)^2 + b) with snr an EMA of the noise sigma (Detector.py:61-63, 81-91,
    177-229; utils.py:334-368) whose memory spans the whole stream.  But the sigma of a due frame is a function of the n
    frames before it only, so the ranks compute the noise sums (S1, S2) of the due frames of their own chunks in a
    cheap first sweep (the sub-ROI of every 2n-th frame's window), all-gather them -- 16 bytes per due frame -- and
    replay the scalar chain on the host (NoiseChain, the same arithmetic as the library's k_noise_finalize); each rank
    then positions its context at its start (mdh_seek) with the exact EMA / threshold state.

The second sweep is the ordinary pipelined detector (mdh_submit_batch / mdh_collect_batch, several passes in flight, frames
read in place when they are resident in HBM) on a context positioned with mdh_seek: it computes the sigmas of its chunk
itself -- the same exact integer sums -- except for the due frames that fall into the first n frames after the seek, whose
window is still filling; those come from the gathered table (mdh_inject_sums).  The Hough stage of the warm-up frames is
skipped (mdh_mute_hough_until).  An engine substituted for the HIP one (CPU tests of the exchange logic) takes the band path
with a single band instead: mdh_band_noise -> mdh_band_pixels (sums from the table) -> mdh_hough_planes.  Results are
identical to one detector running the whole stream either way (tests/test_timeshard.py).
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib
from .bands import HipBandEngine, _BandSetup, _merge_info, _package

ABS_SENSITIVITY = {"high": 3, "normal": 5, "low": 7}          # Detector.py:183
SENSITIVITY_COEF = {"low": (2.0, 4.4), "normal": (1.2, 3.6), "high": (0.9, 3.0)}  # Detector.py:177-182


def due_frames(t: int, n: int, interval: int) -> int:
    """How many frames the sigma evaluated after frame t (1-based) covers; 0 = not due (Detector.py:81-91)."""
    if t > n and t % (interval * n) == 0:
        return n
    if 1 < t <= n:
        return t
    return 0


def due_in(lo: int, hi: int, n: int, interval: int):
    """[(t, k_used)] for the due frames t in (lo, hi], ascending -- due_frames() for a whole pass without a Python step per frame
    (the first sweep of a time window walks a thousand frames per call, and a process that drives several windows does so under
    one interpreter lock)."""
    out = [(t, t) for t in range(max(lo + 1, 2), min(hi, n) + 1)]
    period = interval * n
    first = (max(lo, n) // period + 1) * period
    out.extend((t, n) for t in range(first, hi + 1, period))
    return out


class NoiseChain:
    """EMA of sigma with the warm-up schedule of utils.py:334-368 and the threshold rule of Detector.py:225-229, in the
    operation order of the library's k_noise_finalize / noise_prepare (IEEE doubles, no fused multiply-add on either side)."""

    def __init__(self, n, interval, roi_px, sensitivity, adaptive, init_value):
        self.m0 = 1 - interval / 60
        self.n = n
        self.roi_px = float(roi_px)
        self.adaptive = bool(adaptive)
        self.a, self.b = SENSITIVITY_COEF[sensitivity]
        self.updates = 0
        self.snr = 0.0
        self.sigma = 0.0
        self.thr = ABS_SENSITIVITY[sensitivity] if adaptive else int(init_value)
        self.thr_f = float(self.thr)

    def update(self, k_used: int, s1: int, s2: int):
        m = self.m0
        ramp = self.updates * (1 - self.m0) * self.n
        if ramp < 1:
            u = 1 - ramp
            m = self.m0 * (1 - u * u)
        self.updates += 1
        cnt = float(k_used) * self.roi_px
        mu = float(s1) / cnt
        var = float(s2) / cnt - mu * mu
        if var < 0:
            var = 0.0
        self.sigma = math.sqrt(var)
        self.snr = m * self.snr + (1 - m) * self.sigma
        if self.adaptive and self.snr != 0:
            self.thr_f = self.a * (self.snr * self.snr) + self.b
            self.thr = int(round(self.thr_f))  # half to even

    def state(self):
        return dict(ema_updates=self.updates, snr=self.snr, sigma=self.sigma, thr_float=self.thr_f, thr=self.thr)


def chunk_bounds(total: int, world: int):
    per = -(-total // world)
    return [(min(total, r * per), min(total, (r + 1) * per)) for r in range(world)]


class TimeShard:
    """Rank `rank` of `world`: scan() -> the noise sums of the due frames it owns; after the all-gather, finish(table)
    -> its frames' results.  `read(lo, hi)` returns frames [lo, hi) of the stream as a (hi-lo, H, W) uint8 array."""

    def __init__(self, window_sec, fps, mask, num_cls, cfg, total_frames, read, *, rank=0, world=1, device=0, max_batch=64,
                 engine=HipBandEngine, read_device=None, scan_batch=None):
        """read_device(lo, hi) (optional): device pointer of frames [lo, hi) when the stream is resident in HBM, contiguous.
        scan_batch: frames per call of the first sweep (default max_batch)."""
        self.s = _BandSetup(window_sec, fps, mask, num_cls, cfg, 1)
        self.cfg, self.read, self.total = cfg, read, int(total_frames)
        self.read_device = read_device
        self.pipelined = engine is HipBandEngine
        self.SB = int(scan_batch or max_batch)
        self.rank, self.world, self.B = int(rank), int(world), int(max_batch)
        n = self.s.n
        self.start, self.end = chunk_bounds(self.total, self.world)[self.rank]
        # The warm-up may reach into earlier ranks' chunks (the frames are only read), so a chunk may be shorter than the
        # warm-up itself: every rank derives the same bounds from (total, world) and none can fail alone before a collective.
        self.warm = max(0, self.start - 2 * (n - 1))
        self.engine = engine(self.s.plan, 0, n, cfg, self.s.mask_area, self.SB if self.pipelined else self.B, device)
        self.det = None
        if self.pipelined and self.end > self.start:
            from .detector import M3Detector
            self.det = M3Detector(window_sec, fps, np.asarray(mask), num_cls, cfg, None, device=device, max_batch=self.B)

    def _chain(self):
        b = self.cfg.binary
        return NoiseChain(self.s.n, b.interval, self.s.plan.roi_px, b.sensitivity, b.adaptive_bi_thre, b.init_value)

    def scan(self):
        """First sweep: {t: (k_used, S1, S2)} for the due frames t (1-based) this rank owns."""
        n, itv = self.s.n, self.cfg.binary.interval
        out = {}
        if self.end <= self.start:
            return out
        st0 = self._chain().state()
        self.engine.seek(self.warm, **st0)  # the scalar state plays no part in the sums
        SB = self.SB if self.pipelined else self.B
        for lo in range(self.warm, self.end, SB):
            hi = min(self.end, lo + SB)
            if self.read_device is not None:
                sums = self.engine.noise(self.read_device(lo, hi), K=hi - lo, on_device=True)
            else:
                sums = self.engine.noise(self.read(lo, hi))
            self.engine.skip()
            due = due_in(lo, hi, n, itv)
            assert 2 * len(due) == len(sums)
            for j, (t, k) in enumerate(due):
                if t > self.start:  # owned; its k frames are all behind the warm-up start
                    assert t - k >= self.warm
                    out[t] = (k, int(sums[2 * j]), int(sums[2 * j + 1]))
        return out

    def finish(self, table: dict):
        """Second sweep.  table: the merged {t: (k_used, S1, S2)} of every rank.  Returns [(frame, info, lines, cls_pred)]
        for the frames this rank owns."""
        n, itv = self.s.n, self.cfg.binary.interval
        res = []
        if self.end <= self.start:
            return res
        chain = self._chain()
        for t in sorted(table):
            if t > self.warm:
                break
            chain.update(*table[t])
        if self.pipelined:
            return self._finish_pipelined(table, chain)
        self.engine.seek(self.warm, **chain.state())
        for lo in range(self.warm, self.end, self.B):
            hi = min(self.end, lo + self.B)
            K = hi - lo
            self.engine.noise(self.read(lo, hi))  # stages the pass; its own sums are superseded by the table
            due = [t for t in range(lo + 1, hi + 1) if due_frames(t, n, itv)]
            tot = np.array([v for t in due for v in table[t][1:]], np.int64)
            planes, thr_info = self.engine.pixels(tot)
            if lo < self.start:
                # warm-up frames: their window starts empty, so their binaries are dense garbage; the results are dropped anyway,
                # and replaying HoughLinesP on a million points per frame would dominate the window's run time
                planes[:, :min(K, self.start - lo)] = 0
            hinfo, raw, nms, prob = self.engine.hough_stage(planes, K)
            for j in range(K):
                i = lo + j
                if i >= self.start:
                    info = _merge_info(thr_info[j], hinfo[j])
                    lines, pred = _package(self.s.num_cls, hinfo[j], raw[j], nms[j], prob[j])
                    res.append((i, info, lines, pred))
        return res

    def finish_blocks(self, table: dict):
        """finish() for consumers that work on whole passes (bench, batched collector): one (first_frame, info[K'], rows, nms, prob)
        per pass -- rows = indices (into info) of the frames with segments, nms[r, :info[rows[r]]["n_nms"]] their lineset_nms output,
        prob likewise.  `unpack_block` turns one into the per-frame (lines, cls_pred) of finish().  None when the engine is not
        the HIP one."""
        if not self.pipelined or self.end <= self.start:
            return None
        chain = self._chain()
        for t in sorted(table):
            if t > self.warm:
                break
            chain.update(*table[t])
        return self._finish_pipelined(table, chain, blocks=True)

    def _finish_pipelined(self, table, chain, blocks=False):
        """Second sweep on a full detector context: seek, inject the sums its restarted window cannot produce, mute the
        warm-up frames' Hough stage, then passes in flight like any stream."""
        self._finish_prepare(table, chain)
        for _ in self._finish_steps(blocks):
            pass
        return self._res

    def finish_blocks_steps(self, table: dict, chain=None):
        """finish_blocks() as a generator, for a host that drives SEVERAL windows on one device (bench.py --mode timeshard with one
        process): yields "primed" once the window's pipeline is full (or all its passes are queued), "submitted" when its last
        pass is queued -- the moment to start the next window's passes, so that its fill runs under this window's drain -- and
        ends after the last collect; the records are then in `.blocks`.  The context is positioned (mdh_seek: a device-wide
        synchronisation) HERE, not at the generator's first step: a host that overlaps windows creates all its generators first.
        chain: the scalar chain already replayed up to this window's warm-up start (chains_at: one replay for all the windows of a
        process instead of one per window).  None when the engine is not the HIP one."""
        if not self.pipelined or self.end <= self.start:
            return None
        if chain is None:
            chain = self._chain()
            for t in sorted(table):
                if t > self.warm:
                    break
                chain.update(*table[t])
        self._finish_prepare(table, chain)
        return self._finish_steps(True)

    @property
    def blocks(self):
        return self._res

    def _finish_prepare(self, table, chain):
        n = self.s.n
        ctx = self.det._ctx
        ctx.seek(self.warm, **chain.state())
        if self.warm > 0:
            for t in range(self.warm + 1, self.warm + n):  # fewer than n frames since the seek: the window is still filling
                if t in table:
                    ctx.inject_sums(t, table[t][1], table[t][2])
        ctx.mute_hough_until(self.start)

    def _finish_steps(self, blocks):
        ctx = self.det._ctx
        B = self.B
        bufs = [(np.zeros(B, _lib.FRAME_INFO_DTYPE), np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32),
                 np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32), np.zeros((B, _lib.MDH_MAX_LINES), np.float64))]
        queue, res = [], []
        self._res = res

        def collect():
            lo, K, _keep = queue.pop(0)
            info, raw, nms, prob = bufs[0]
            ctx.collect_into(info, raw, nms, prob)
            first = max(0, self.start - lo)  # frames before it are warm-up: dropped
            if blocks:
                # one record per pass, arrays only (what a batched collector takes): the per-frame scalars, and the NMS segments /
                # non-line probabilities of the frames that have any, cut to the widest of them
                cnt = info["n_nms"][first:K]
                rows = np.flatnonzero(cnt > 0) + first
                cmax = int(cnt.max()) if len(rows) else 0
                res.append((lo + first, info[first:K].copy(), rows - first, nms[rows, :cmax].copy(), prob[rows, :cmax].copy()))
                return
            for j in range(first, K):
                lines, pred = _package(self.s.num_cls, info[j], raw[j], nms[j], prob[j])
                res.append((lo + j, info[j].copy(), lines, pred))

        starts = list(range(self.warm, self.end, B))
        primed = False
        for k, lo in enumerate(starts):
            hi = min(self.end, lo + B)
            if ctx.in_flight >= ctx.max_in_flight:
                if not primed:
                    primed = True
                    yield "primed"
                collect()
            if self.read_device is not None:
                ctx.submit(self.read_device(lo, hi), hi - lo, True)
                queue.append((lo, hi - lo, None))
            else:
                frames = np.ascontiguousarray(self.read(lo, hi))
                ctx.submit(frames.ctypes.data, hi - lo, False)
                queue.append((lo, hi - lo, frames))  # the frames must stay valid until the pass is collected
        if not primed:
            yield "primed"
        yield "submitted"
        while queue:
            collect()

    def close(self):
        self.engine.close()
        if self.det is not None:
            self.det.close()
            self.det = None


def chains_at(shards, table: dict):
    """The scalar chain replayed ONCE over the gathered table, a copy taken where each shard's warm-up starts: [NoiseChain] in the
    order of `shards` (for a process that drives several windows: every TimeShard replaying the chain from frame 0 by itself is
    O(windows x due frames) of interpreter time inside the timed region).  The per-update arithmetic that does not depend on the
    chain -- sigma from the sums, the momentum schedule -- is done on arrays (IEEE doubles, the same operations in the same order
    as NoiseChain.update: element for element the same values); only the recurrence itself is a loop."""
    import copy
    proto = shards[0]._chain()
    keys = sorted(table)
    if not keys:
        return [copy.copy(proto) for _ in shards]
    k_used = np.array([table[t][0] for t in keys], np.float64)
    s1 = np.array([table[t][1] for t in keys], np.int64).astype(np.float64)
    s2 = np.array([table[t][2] for t in keys], np.int64).astype(np.float64)
    cnt = k_used * proto.roi_px
    mu = s1 / cnt
    var = s2 / cnt - mu * mu
    var[var < 0] = 0.0
    sig = np.sqrt(var)
    upd = np.arange(len(keys), dtype=np.float64)
    ramp = upd * (1 - proto.m0) * proto.n
    u = 1 - ramp
    m = np.where(ramp < 1, proto.m0 * (1 - u * u), proto.m0)
    om = 1 - m
    ml, oml, sl = m.tolist(), om.tolist(), sig.tolist()
    # where each shard's chain stops: the number of table entries with t <= warm
    stops = np.searchsorted(np.asarray(keys), [sh.warm for sh in shards], side="right").tolist()
    order = sorted(range(len(shards)), key=lambda q: stops[q])
    out = [None] * len(shards)
    snr, j, last_nz = 0.0, 0, None  # last_nz: snr after the last update that left it non-zero (what the threshold follows)
    for q in order:
        while j < stops[q]:
            snr = ml[j] * snr + oml[j] * sl[j]
            if snr != 0:
                last_nz = snr
            j += 1
        ch = copy.copy(proto)
        ch.updates, ch.snr = j, snr
        ch.sigma = sl[j - 1] if j else 0.0
        if ch.adaptive and last_nz is not None:
            ch.thr_f = ch.a * (last_nz * last_nz) + ch.b
            ch.thr = int(round(ch.thr_f))  # half to even
        out[q] = ch
    return out


def unpack_block(block, num_cls):
    """[(frame, info, lines, cls_pred)] of one finish_blocks() record, as finish() returns them."""
    f0, info, rows, nms, prob = block
    out = []
    where = {int(r): k for k, r in enumerate(rows)}
    for j in range(len(info)):
        if j in where:
            k, c = where[j], int(info[j]["n_nms"])
            pred = np.zeros((c, num_cls))
            pred[:, -1] = prob[k, :c]
            pred[:, 0] = 1 - prob[k, :c]
            out.append((f0 + j, info[j], nms[k, :c].copy(), pred))
        else:
            out.append((f0 + j, info[j], np.array([]), np.zeros((0, num_cls))))
    return out


def run_time_sharded(window_sec, fps, mask, num_cls, cfg, total_frames, read, *, group=None, rank=0, world=1, device=0,
                     max_batch=64, engine=HipBandEngine):
    """All of it for one rank: scan, all-gather of the sums table, finish, gather of the results to rank 0.
    Returns (results, info) on rank 0 -- results[i] = (lines, cls_pred) as M3Detector.detect gives them -- else None.
    group: `dist.Group` (RCCL all-gather when every rank has its own device, the rendezvous store otherwise) or anything with
    its all_gather_object / gather_object."""
    if world > 1 and group is None:
        raise ValueError("world > 1 needs a group (metdetpy_amd.dist.init_from_env)")
    shard = TimeShard(window_sec, fps, mask, num_cls, cfg, total_frames, read, rank=rank, world=world, device=device,
                      max_batch=max_batch, engine=engine)
    mine = shard.scan()
    tables = [mine] if group is None else group.all_gather_object(mine)  # the threshold table: 16 bytes of sums per due frame
    table = {}
    for part in tables:
        table.update(part)
    part = shard.finish(table)
    shard.close()
    gathered = [part] if group is None else group.gather_object(part, 0)  # the final detection gather
    if rank != 0:
        return None
    info = np.zeros(total_frames, _lib.FRAME_INFO_DTYPE)
    results = [None] * total_frames
    for chunk in gathered:
        for i, fi, lines, pred in chunk:
            info[i] = fi
            results[i] = (lines, pred)
    return results, info
