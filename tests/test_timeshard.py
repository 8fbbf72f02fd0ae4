"""One stream cut into time windows across ranks (metdetpy_amd/timeshard.py; SURVEY 8e "time windows of one stream").

CPU: the host replay of the noise EMA / threshold chain against the oracle's series, the two-sweep scheme with an engine
made of the oracle's functions (ranks simulated in-process, and two real processes over gloo) -- identical to one oracle
detector running the whole stream.  GPU (-m gpu): the same with the HIP contexts, against the whole-stream HIP detector."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from metdetpy_amd.timeshard import NoiseChain, TimeShard, chunk_bounds, due_frames, run_time_sharded
from metdetpy_amd.utils import select_std_roi
from oracle import m3_oracle as mo
from tests.band_oracle_engine import OracleBandEngine
from tests.helpers import binary_cfg, frames_normal
from tests.mp import run_ranks
from tests.test_bands import check_against, make_case, whole_frame_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chunk_bounds_and_due_frames():
    assert chunk_bounds(100, 3) == [(0, 34), (34, 68), (68, 100)]
    assert chunk_bounds(5, 8)[-1] == (5, 5)
    n, itv = 4, 2
    assert [due_frames(t, n, itv) for t in range(1, 18)] == [0, 2, 3, 4, 0, 0, 0, 4, 0, 0, 0, 0, 0, 0, 0, 4, 0]


@pytest.mark.parametrize("sens,sigma", [("normal", 2.5), ("high", 1.0), ("low", 6.0)])
def test_noise_chain_replays_the_oracle_series(sens, sigma):
    H, W, n = 60, 100, 5
    mask = np.ones((H, W), np.uint8)
    cfg = binary_cfg(sens)
    frames = frames_normal(11, 90, (H, W), 30, sigma)
    orc = mo.M3Oracle(n, mask, sensitivity=sens)
    r = select_std_roi(mask, 0.1)
    chain = NoiseChain(n, 2, (r[2] - r[0]) * (r[3] - r[1]), sens, True, 7)
    changes = 0
    for t, f in enumerate(frames, 1):
        orc.update(f)
        k = due_frames(t, n, cfg.binary.interval)
        if k:
            s1, s2, _ = mo.noise_sums(orc.stack.sub.ring, orc.stack.sub.mean, k)
            before = chain.thr
            chain.update(k, s1, s2)
            changes += chain.thr != before
        assert chain.thr == orc.bi_threshold, t
        assert chain.snr == pytest.approx(orc.stack.snr, rel=1e-12, abs=0)
    assert changes > 0


def simulate(world, frames, n, mask, cfg, B, engine, resident=False):
    total = len(frames)
    read = lambda lo, hi: frames[lo:hi]
    shards = [TimeShard(n / 10, 10, mask, 10, cfg, total, read, rank=r, world=world, max_batch=B, engine=engine)
              for r in range(world)]
    pool = None
    if resident:  # the stream lives in HBM: both sweeps read it in place
        ctx = shards[0].engine.band
        frames = np.ascontiguousarray(frames)
        pool = ctx.device_alloc(frames.nbytes)
        ctx.device_upload(pool, frames)
        fb = frames[0].nbytes
        for s in shards:
            s.read_device = lambda lo, hi: pool + lo * fb
    table = {}
    for s in shards:
        part = s.scan()
        assert not (set(part) & set(table))
        table.update(part)
    assert sorted(table) == [t for t in range(1, total + 1) if due_frames(t, n, cfg.binary.interval)]
    out = []
    for s in shards:
        out += s.finish(table)
    if pool is not None:
        shards[0].engine.band.device_free(pool)
    for s in shards:
        s.close()
    assert [o[0] for o in out] == list(range(total))
    return out


@pytest.mark.parametrize("kind,world,H,W,n,total,B", [("sky", 2, 96, 160, 4, 60, 16), ("sky", 3, 90, 128, 6, 96, 11),
                                                       ("busy", 2, 64, 96, 5, 40, 7), ("sky", 1, 64, 128, 4, 30, 8)])
def test_simulated_ranks_equal_whole_stream_oracle(kind, world, H, W, n, total, B):
    mask = np.ones((H, W), np.uint8)
    frames, cfg = make_case(kind, H, W, n, total, 5 + world)
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert sum(w["n_raw"] for w in want) > 0 and len({w["thr"] for w in want}) > (1 if kind == "sky" else 0)
    out = simulate(world, frames, n, mask, cfg, B, OracleBandEngine)
    check_against([(o[2], o[3]) for o in out], [o[1] for o in out], want)


TS_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
from metdetpy_amd.timeshard import run_time_sharded
from tests.band_oracle_engine import OracleBandEngine
from tests.gloo_group import make_group
from tests.test_bands import make_case
use_hip, kind = sys.argv[2] == "hip", sys.argv[3]
H, W, n, total, B = [int(v) for v in sys.argv[4:9]]
group = make_group(kind)
rank, world = group.rank, group.world
mask = np.ones((H, W), np.uint8)
frames, cfg = make_case("sky", H, W, n, total, 31)
kw = {} if use_hip else dict(engine=OracleBandEngine)
res = run_time_sharded(n / 10, 10, mask, 10, cfg, total, lambda lo, hi: frames[lo:hi], group=group, rank=rank, world=world,
                       max_batch=B, **kw)
if rank == 0:
    results, info = res
    print(json.dumps([dict(lines=np.asarray(l).reshape(-1, 4).tolist(), pred=np.asarray(p).tolist(), thr=int(fi["bi_threshold"]),
                           lit=int(fi["lit_count"]), n_raw=int(fi["n_raw"])) for (l, p), fi in zip(results, info)]))
else:
    assert res is None
group.close()
"""


def run_ts_ranks(tmp_path, mode, kind, world, H, W, n, total, B):
    return run_ranks(tmp_path, "ts_worker", TS_WORKER, world, args=(mode, kind, H, W, n, total, B))


def compare(got, H, W, n, total):
    mask = np.ones((H, W), np.uint8)
    frames, cfg = make_case("sky", H, W, n, total, 31)
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert len(got) == total and sum(w["n_raw"] for w in want) > 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert (g["thr"], g["lit"], g["n_raw"]) == (w["thr"], w["lit"], w["n_raw"]), i
        assert np.array_equal(np.array(g["lines"], np.int32).reshape(-1, 4), w["lines"]), i
        assert np.array_equal(np.array(g["pred"]).reshape(-1, 10), w["pred"]), i


@pytest.mark.parametrize("kind", ["store", "gloo"])
def test_two_process_time_shards_equal_whole_stream_oracle(tmp_path, kind):
    compare(run_ts_ranks(tmp_path, "oracle", kind, 2, 96, 160, 5, 70, 12), 96, 160, 5, 70)


@pytest.mark.parametrize("kind", ["store", "gloo"])
def test_eight_process_time_windows_with_the_all_gathered_table(tmp_path, kind):
    """north_star's sharding at its world size: 8 windows of one stream, the (S1, S2) table of the due frames all-gathered, every
    rank replaying the scalar chain up to its own start, results gathered to rank 0 -- against ONE oracle over the whole stream.
    136 frames over 8 ranks = 17 per window: shorter than the 2(n-1) = 10-frame warm-up plus a pass, so windows reach into their
    predecessors' frames, and the due frames (every 2n = 12) fall into different windows."""
    compare(run_ts_ranks(tmp_path, "oracle", kind, 8, 96, 160, 6, 136, 7), 96, 160, 6, 136)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kind,world,H,W,n,total,B", [("sky", 2, 270, 480, 8, 120, 32), ("sky", 3, 135, 250, 5, 100, 16),
                                                       ("busy", 2, 128, 192, 4, 48, 8), ("sky", 4, 272, 448, 8, 200, 64)])
def test_hip_time_shards_equal_whole_stream(kind, world, H, W, n, total, B):
    from metdetpy_amd import M3Detector
    from metdetpy_amd.bands import HipBandEngine
    mask = np.ones((H, W), np.uint8)
    frames, cfg = make_case(kind, H, W, n, total, 5 + world)
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert sum(w["n_raw"] for w in want) > 0
    out = simulate(world, frames, n, mask, cfg, B, HipBandEngine)
    check_against([(o[2], o[3]) for o in out], [o[1] for o in out], want)
    out_dev = simulate(world, frames, n, mask, cfg, B, HipBandEngine, resident=True)
    check_against([(o[2], o[3]) for o in out_dev], [o[1] for o in out_dev], want)
    whole = M3Detector(n / 10, 10, mask, 10, cfg, None, max_batch=B)
    _, winfo = whole.detect_batch(frames)
    whole.close()
    for i, o in enumerate(out):
        for f in ("n_raw", "n_nms", "lit_count", "bi_threshold", "bi_threshold_float", "snr", "sigma", "dst_sum", "hough_gap"):
            assert o[1][f] == winfo[i][f], (i, f, o[1][f], winfo[i][f])  # the host chain replays the device chain bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["store", "gloo"])
def test_two_process_hip_time_shards(tmp_path, kind):
    compare(run_ts_ranks(tmp_path, "hip", kind, 2, 270, 448, 8, 150, 32), 270, 448, 8, 150)


@pytest.mark.gpu
def test_eight_process_hip_time_windows_on_one_gpu(tmp_path):
    """8 windows, eight processes sharing the box's one GPU (the pipelined detector of every window on its own context)."""
    compare(run_ts_ranks(tmp_path, "hip", "store", 8, 270, 448, 8, 400, 32), 270, 448, 8, 400)


@pytest.mark.gpu
def test_seek_errors_and_restart():
    """mdh_seek on a full context: rewinding to 0 reproduces the first run; seeking with a pass staged is refused."""
    from metdetpy_amd import M3Detector, _lib
    H, W, n = 96, 128, 4
    mask = np.ones((H, W), np.uint8)
    frames, cfg = make_case("sky", H, W, n, 24, 3)
    det = M3Detector(n / 10, 10, mask, 10, cfg, None, max_batch=8)
    first, info1 = det.detect_batch(frames)
    det._ctx.seek(0, thr=5, thr_float=5.0)
    again, info2 = det.detect_batch(frames)
    for a, b in zip(first, again):
        assert np.array_equal(np.asarray(a[0]).reshape(-1, 4), np.asarray(b[0]).reshape(-1, 4))
    assert np.array_equal(info1, info2)
    det._ctx.update(frames[0])
    with pytest.raises(_lib.MetDetHipError, match="in flight"):
        det._ctx.seek(0)
    det.close()


@pytest.mark.gpu
def test_inject_sums_only_for_due_frames():
    """mdh_inject_sums: only a frame whose sigma is due (Detector.py:82,87: t > n and t % (interval * n) == 0, or 1 < t <= n) ever
    looks its entry up, so anything else is refused instead of sitting in the 64-entry table for good; a second entry for the same
    frame replaces the first; frames already submitted are refused."""
    from metdetpy_amd import M3Detector, _lib
    H, W, n = 64, 128, 4
    mask = np.ones((H, W), np.uint8)
    frames, cfg = make_case("sky", H, W, n, 8, 3)   # interval 2: due at t = 2, 3, 4 (warm-up) and 8, 16, 24, ...
    det = M3Detector(n / 10, 10, mask, 10, cfg, None, max_batch=8)
    ctx = det._ctx
    for t in (5, 6, 7, 9, 15, 17):
        with pytest.raises(_lib.MetDetHipError, match="no noise sigma is due"):
            ctx.inject_sums(t, 1, 2)
    for _ in range(70):              # far more calls than the table holds: the same frame again replaces its entry
        ctx.inject_sums(8, 10, 20)
    ctx.inject_sums(16, 1, 2)
    ctx.inject_sums(3, 1, 2)
    det.detect_batch(frames)         # frames 1..8 consumed: their entries are used up
    with pytest.raises(_lib.MetDetHipError, match="already been submitted"):
        ctx.inject_sums(8, 1, 2)
    det.close()


@pytest.mark.gpu
def test_seek_then_resident_frames_at_once():
    """mdh_seek wipes the tagged binaries with device-side fills; the passes submitted right behind it -- frames resident in HBM,
    so nothing delays the first kernels -- must not race those fills (they did: streak frames lost their lit pixels)."""
    from metdetpy_amd import M3Detector
    H, W, n, K = 540, 960, 8, 256
    mask = np.ones((H, W), np.uint8)
    frames, cfg = make_case("sky", H, W, n, K, 21)
    frames = np.ascontiguousarray(frames)
    det = M3Detector(n / 10, 10, mask, 10, cfg, None, max_batch=K)
    ctx = det._ctx
    pool = ctx.device_alloc(frames.nbytes)
    ctx.device_upload(pool, frames)
    first = ctx.process_batch(pool, K=K, on_device=True)
    assert int(first[0]["n_raw"].sum()) > 0
    for _ in range(3):
        ctx.seek(0, thr=5, thr_float=5.0)
        again = ctx.process_batch(pool, K=K, on_device=True)
        assert np.array_equal(first[0], again[0])
        assert np.array_equal(first[1], again[1])
    ctx.device_free(pool)
    det.close()


@pytest.mark.gpu
def test_band_skip_between_real_passes():
    """A skipped pass in the middle of a stream (its frames enter the window, no pixel stages run) must leave the slots it
    touches clean: with a fixed threshold, every pass that starts a full pass after a skip equals the context that skipped
    nothing (the pass right after a skip may differ in its first n-1 frames: the dynamic mask has no binaries for the
    skipped frames)."""
    from metdetpy_amd.bands import BandPlan, HipBandEngine
    H, W, n, B = 128, 192, 4, 8
    mask = np.ones((H, W), np.uint8)
    frames, cfg = make_case("busy", H, W, n, 14 * B, 9)
    assert not cfg.binary.adaptive_bi_thre
    plan = BandPlan(H, W, 1, select_std_roi(mask, cfg.binary.area))
    full = HipBandEngine(plan, 0, n, cfg, float(mask.sum()), B, 0)
    part = HipBandEngine(plan, 0, n, cfg, float(mask.sum()), B, 0)
    skipped = {2, 3, 7, 12}
    compared = lit = 0
    for p in range(14):
        chunk = frames[p * B:(p + 1) * B]
        tot = full.noise(chunk)
        planes_f, _ = full.pixels(tot)
        assert np.array_equal(tot, part.noise(chunk))  # the noise window saw the same frames
        if p in skipped:
            part.skip()
            continue
        planes_p, _ = part.pixels(tot)
        if p - 1 not in skipped:
            assert np.array_equal(planes_f, planes_p), p
            compared += 1
            lit += int(np.count_nonzero(planes_f))
    assert compared >= 6 and lit > 0
    full.close()
    part.close()


def test_chains_at_equals_the_sequential_replay():
    """timeshard.chains_at (one replay of the scalar chain for all windows of a process, sigma and momenta on arrays) against
    NoiseChain.update entry by entry: identical states -- doubles compared exactly -- wherever a window's warm-up starts; and due_in
    against due_frames frame by frame."""
    from metdetpy_amd.timeshard import NoiseChain, chains_at, due_frames, due_in

    class Shard:
        def __init__(self, warm, adaptive):
            self.warm, self.adaptive = warm, adaptive

        def _chain(self):
            return NoiseChain(8, 2, 20736.0, "normal", self.adaptive, 7)

    rng = np.random.default_rng(1)
    for trial in range(60):
        adaptive = bool(trial % 2)
        ts = sorted(rng.choice(np.arange(2, 5000), size=int(rng.integers(0, 300)), replace=False).tolist())
        table = {int(t): (8 if t > 8 else int(t), int(rng.integers(-10**6, 10**6)) if trial % 5 else 0,
                          int(rng.integers(0, 10**9)) if trial % 5 else 0) for t in ts}
        shards = [Shard(int(w), adaptive) for w in rng.integers(0, 5200, size=6)]
        for sh, got in zip(shards, chains_at(shards, table)):
            ref = sh._chain()
            for t in sorted(table):
                if t > sh.warm:
                    break
                ref.update(*table[t])
            assert ref.state() == got.state(), (trial, sh.warm)
    for _ in range(300):
        n, itv, lo = int(rng.integers(2, 20)), int(rng.integers(1, 5)), int(rng.integers(0, 300))
        hi = lo + int(rng.integers(0, 200))
        assert due_in(lo, hi, n, itv) == [(t, due_frames(t, n, itv)) for t in range(lo + 1, hi + 1) if due_frames(t, n, itv)]
