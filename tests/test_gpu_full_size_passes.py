"""Parity at the sizes that are TIMED (run on the MI355X box: pytest -m gpu).

bench.py's throughput figures come from full-size passes resubmitted over and over: 4000 frames of 1920x1080 per pass (time chunk
boundaries near frames 1334 and 2668, history carried from pass to pass, result slots reused), 1000 frames of 3840x2160 at n = 16
under mask-east, 500 frames of 7680x4320 at m3det_high in either output format of the window kernel.  The other parity tests run
fresh detectors on <= 256 frames; these run the timed shapes themselves -- the same resident pool, the same submit / collect calls
-- and compare EVERY frame of every pass with the C port of the oracle (oracle/metdet_oracle.c): threshold, lit count, the raw
Hough segments in discovery order, the lineset_nms output.

Selection (round 6): `pytest -m gpu` runs one full-size pass per shape -- 1080p twice on one detector, 4K, 8K in its production
(dense) format, and the long windows n = 25 and 32 through bench.py's own timed-region check; `-m "gpu and slow"` adds the
8K pass in the sparse format and the long windows n = 17 and 30 (tests/conftest.py leaves `slow` tests out unless the marker
expression names them: the whole default selection must stay well inside the driver's 1200 s step).
"""
import os

import numpy as np
import pytest

import bench
from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.utils import select_std_roi
from oracle import c_oracle as co
from tests.helpers import binary_cfg

pytestmark = pytest.mark.gpu


def oracle_rows(frames, idx, n, mask, sensitivity):
    """[(thr, lit, n_raw, raw, nms)] of the stream frames[idx[0]], frames[idx[1]], ... from ONE oracle detector."""
    co.lib().om3_set_threads(min(16, len(os.sched_getaffinity(0))))
    orc = co.CM3(n, mask, select_std_roi(mask, 0.1), sensitivity=sensitivity)
    rows = []
    for i in idx:
        orc.update(frames[i])
        raw = orc.detect()
        nms = co.lineset_nms(raw)[0] if 0 < orc.lines_num <= 500 else None
        rows.append((orc.state()[0], orc.lit_count, orc.lines_num, raw[:512].copy(), nms))
    return rows


def run_passes(wlname, K, passes, uniq, sensitivity="normal", mask=None):
    """`passes` passes of K frames, submitted back to back on ONE detector (the second continues the stream of the first: the pool
    holds whole periods), from a pool resident in HBM, as bench.py's timed loop does.  Returns (frames, idx, [per-pass arrays])."""
    wl = bench.WORKLOADS[wlname]
    H, W, n = wl["H"], wl["W"], wl["n"]
    mask = np.ones((H, W), np.uint8) if mask is None else mask
    frames = bench.make_stream_frames(wl, 0, mask, K, uniq)
    assert K % len(frames) == 0, "the pool must hold whole periods of the stream, or a resubmitted pass would start on a scene cut"
    det = M3Detector(1, n, mask, 10, binary_cfg(sensitivity), None, max_batch=K)
    ctx = det._ctx
    pool = ctx.device_alloc(K * H * W)
    for s in range(0, K, len(frames)):
        ctx.device_upload(pool + s * H * W, frames)
    out = []
    for _ in range(passes):
        ctx.submit(pool, K, True)
    for _ in range(passes):
        info = np.zeros(K, _lib.FRAME_INFO_DTYPE)
        raw = np.zeros((K, _lib.MDH_MAX_LINES, 4), np.int32)
        nms = np.zeros((K, _lib.MDH_MAX_LINES, 4), np.int32)
        prob = np.zeros((K, _lib.MDH_MAX_LINES), np.float64)
        assert ctx.collect_into(info, raw, nms, prob) == K
        out.append((info, raw, nms))
    dense = ctx.dense_passes
    ctx.device_free(pool)
    det.close()
    return frames, np.arange(passes * K) % len(frames), out, dense


def compare(rows, out, K):
    segs = 0
    for p, (info, raw, nms) in enumerate(out):
        want = rows[p * K:(p + 1) * K]
        thr = np.array([w[0] for w in want])
        lit = np.array([w[1] for w in want])
        nr = np.array([w[2] for w in want])
        for name, got, exp in (("bi_threshold", info["bi_threshold"], thr), ("lit_count", info["lit_count"], lit), ("n_raw", info["n_raw"], nr)):
            bad = np.flatnonzero(got != exp)
            assert len(bad) == 0, f"pass {p}: {name} differs at frames {bad[:8]} (of {len(bad)}): {got[bad[:4]]} vs {exp[bad[:4]]}"
        for i in np.flatnonzero(nr):
            w = want[i]
            assert np.array_equal(raw[i][:min(w[2], 512)], w[3]), f"pass {p}, frame {i}: raw segments"
            if w[4] is not None:
                assert int(info[i]["n_nms"]) == len(w[4]) and np.array_equal(nms[i][:len(w[4])], w[4]), f"pass {p}, frame {i}: NMS"
            segs += w[2]
    return segs


def test_1080p_pass_of_4000_frames_twice():
    """The headline shape: K = 4000 at 1920x1080, n = 8 -- three time chunks per launch -- submitted twice, so the second pass
    starts from the first one's window history, ring of binaries and scalar state.  The oracle runs the first pass and the first
    four periods of the second (the hand-over); the rest of the second pass is the same periodic stream at a settled threshold, and
    is held frame by frame to the period the oracle did check."""
    K, P = 4000, 250
    frames, idx, out, _ = run_passes("C2", K, 2, P)
    rows = oracle_rows(frames, idx[:K + 4 * P], 8, np.ones(frames.shape[1:], np.uint8), "normal")
    for a, b in zip(rows[K + 2 * P:K + 3 * P], rows[K + 3 * P:K + 4 * P]):  # the oracle itself repeats from period to period there
        assert a[:3] == b[:3] and np.array_equal(a[3], b[3])
    info2, raw2, nms2 = out[1]
    segs = compare(rows[:K], out[:1], K) + compare(rows[K:], [(info2[:4 * P], raw2[:4 * P], nms2[:4 * P])], 4 * P)
    for s in range(4 * P, K, P):  # every later period of the second pass == its fourth period, which the oracle checked
        for name in ("bi_threshold", "lit_count", "n_raw", "n_nms"):
            assert np.array_equal(info2[name][s:s + P], info2[name][3 * P:4 * P]), (s, name)
        live_raw = np.arange(raw2.shape[1])[None, :] < info2["n_raw"][s:s + P, None]   # (rows past a frame's count are not written)
        live_nms = np.arange(nms2.shape[1])[None, :] < info2["n_nms"][s:s + P, None]
        assert np.array_equal(raw2[s:s + P][live_raw], raw2[3 * P:4 * P][live_raw]), s
        assert np.array_equal(nms2[s:s + P][live_nms], nms2[3 * P:4 * P][live_nms]), s
    assert segs > 16 * 400 + 4 * 400, segs  # (550 segments per 250-frame period once the threshold has settled)
    # steady state: from the third period on every period repeats the same tallies -- what bench.py's timed region relies on
    per = [(int(out[1][0]["n_raw"][s:s + P].sum()), int(out[1][0]["lit_count"][s:s + P].sum())) for s in range(0, K, P)]
    assert len(set(per)) == 1, per


def test_4k_pass_of_1000_frames_window16_mask():
    K = 1000
    mask = bench.load_mask(2160, 3840)
    frames, idx, out, _ = run_passes("C3", K, 1, 100, mask=mask)
    rows = oracle_rows(frames, idx, 16, mask, "normal")
    assert compare(rows, out, K) > 500


@pytest.mark.parametrize("dense", [pytest.param("0", marks=pytest.mark.slow), "1"])  # (8K's production format is the dense one)
def test_8k_pass_of_500_frames_both_output_formats(dense, monkeypatch):
    """7680x4320 at m3det_high: the window kernel's two output formats (MDH_DENSE pins one) over a full 500-frame pass."""
    monkeypatch.setenv("MDH_DENSE", dense)
    K = 500
    frames, idx, out, dense_passes = run_passes("C5", K, 1, 50, sensitivity="high")
    assert dense_passes == int(dense)
    global _rows_8k
    try:
        rows = _rows_8k
    except NameError:
        rows = _rows_8k = oracle_rows(frames, idx, 8, np.ones(frames.shape[1:], np.uint8), "high")
    assert compare(rows, out, K) > 100


@pytest.mark.parametrize("n", [pytest.param(17, marks=pytest.mark.slow), 25, pytest.param(30, marks=pytest.mark.slow), 32])
def test_long_window_full_size_passes_through_the_bench(n):
    """Windows of 17 .. 32 frames (the reference's default window_sec 1 gives n = 25 / 30, Detector.py:200; 8-pixels-per-lane window
    kernels since round 6): full-size 1080p passes of 4000 frames -- two time chunks per launch, the second pass continuing the
    first's window -- through bench.py itself: the first periods frame by frame against the oracle AND the timed region's tallies
    (raw segments, NMS segments, lit pixels of every pass) against the oracle's tallies carried through the periodic stream."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--window", str(n), "--unique", "100", "--steps", "2", "--warmup", "1",
                        "--passes-per-step", "3", "--no-secondary", "--no-pmc", "--cpu-seconds", "0", "--parity-frames", "128", "--standalone-passes", "2"],
                       capture_output=True, text=True, timeout=580, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["window"] == n and d["config"]["frames_per_pass"] == 4000
    assert d["parity"]["verdict"] == "identical" and d["parity"]["frames_checked"] >= 128, d["parity"]
    assert d["parity"]["timed_region"]["verdict"] == "identical", d["parity"]["timed_region"]
    assert d["detections"]["raw_segments"] > 0 and 0.2 < d["roofline"]["frac_isolated"] < 1
