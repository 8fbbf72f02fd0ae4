"""HIP path vs the CPU oracle, through the C-ABI (run on the MI355X box: pytest -m gpu).

Bar: bit-exact -- binary images, lit counts, threshold series, raw Hough segments (discovery order), NMS output.
The noise EMA is float64 on both sides but summed differently (exact int64 sums on the GPU vs np.std): rtol 1e-12.
"""
import os

import numpy as np
import pytest

from metdetpy_amd import ClassicDetector, M3Detector
from metdetpy_amd.synth import SkyStream
from oracle import c_oracle as co
from oracle import m3_oracle as mo
from tests.helpers import binary_cfg, busy_frames, frames_normal, oracle_kwargs

pytestmark = pytest.mark.gpu


def run_pair(frames, n, mask, cfg, *, batch=None, check_act=True, max_batch=64, pinned=False):
    """Feed the same frames to the GPU detector and the oracle; assert parity frame by frame.  pinned: the per-frame protocol
    gets its frames in pinned host memory (mdh_host_alloc), which the window kernel then reads itself (no DMA copy)."""
    H, W = mask.shape
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=max_batch)
    pool = None
    if pinned:
        import ctypes
        arr = np.ascontiguousarray(np.stack(frames))
        pool = det._ctx.host_alloc(arr.nbytes)
        ctypes.memmove(pool, arr.ctypes.data, arr.nbytes)
        frames = np.ctypeslib.as_array((ctypes.c_uint8 * arr.nbytes).from_address(pool)).reshape(arr.shape)
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    assert tuple(det.stack.std_roi) == tuple(orc.stack.std_roi)
    total_lines = 0

    def compare(t, lines, cls_pred, info, dst, act):
        nonlocal total_lines
        o_lines, o_pred = orc._last
        assert int(info["bi_threshold"]) == orc.bi_threshold, f"frame {t}: threshold"
        assert float(info["snr"]) == pytest.approx(float(orc.stack.snr), rel=1e-12, abs=1e-300), f"frame {t}: snr"
        if act is not None:
            assert np.array_equal(act, orc.act), f"frame {t}: pre-mask binary differs at {np.argwhere(act != orc.act)[:5]}"
        assert np.array_equal(dst, orc.dst), f"frame {t}: dst differs at {np.argwhere(dst != orc.dst)[:5]}"
        assert int(info["lit_count"]) == orc.lit_count
        assert float(info["dst_sum"]) == orc.dst_sum
        assert float(info["hough_gap"]) == orc.gap
        assert int(info["n_raw"]) == orc.lines_num, f"frame {t}: raw segment count"
        assert np.array_equal(np.asarray(lines).reshape(-1, 4), np.asarray(o_lines).reshape(-1, 4)), f"frame {t}: NMS lines"
        assert np.array_equal(cls_pred, o_pred), f"frame {t}: cls_pred"
        total_lines += len(o_lines)

    if batch is None:
        for t, f in enumerate(frames):
            det.update(f)
            lines, pred = det.detect()
            orc.update(f)
            orc._last = orc.detect()
            info = det._ctx._info1[0]
            assert np.array_equal(det.linesp_ext.reshape(-1, 4),
                                  (orc.raw_lines if orc.lines_num <= 500 else orc.raw_lines[:0]).reshape(-1, 4)), f"frame {t}: raw"
            compare(t, lines, pred, info, det.dst, det._ctx.get_binary(-1, 1) if check_act else None)
    else:
        t = 0
        arr = np.stack(frames)
        while t < len(arr):
            chunk = arr[t:t + batch]
            res, info = det.detect_batch(chunk)
            for i in range(len(chunk)):
                orc.update(chunk[i])
                orc._last = orc.detect()
                passes_before = (i // max_batch) * max_batch
                in_last_pass = i >= ((len(chunk) - 1) // max_batch) * max_batch
                dst = det._ctx.get_binary(i - passes_before, 0) if in_last_pass else None
                if dst is None:  # only the last internal pass is retained on the device
                    assert int(info[i]["lit_count"]) == orc.lit_count
                    dst = orc.dst
                compare(t + i, res[i][0], res[i][1], info[i], dst, None)
            t += len(chunk)
    if pool is not None:
        del frames
        det._ctx.host_free(pool)
    det.close()
    return total_lines


@pytest.mark.parametrize("shape,n", [((128, 256), 8), ((135, 192), 4), ((96, 128), 16), ((64, 64), 2), ((72, 192), 25), ((40, 128), 32)])
def test_m3_per_frame_from_pinned_memory(shape, n):
    """The per-frame protocol's fast path (the window kernel reads the caller's pinned buffer and the pointer table from
    host memory, no DMA / prepare / noise commands when no sigma is due; results announced through the polled word): same
    bar as every other path, dst and the pre-mask binary included, across several sigma-due frames (interval * n)."""
    mask = np.ones(shape, np.uint8)
    s = SkyStream(*shape, 31 + n, speed=5, streak_period=11, streak_life=7)
    total = run_pair([s.next() for _ in range(5 * n + 40)], n, mask, binary_cfg("normal"), pinned=True)
    assert total > 0 or n == 2  # (a two-frame window keeps too little of a streak for a segment)
    # widths that are not a multiple of 64 take the general path from the same pinned buffers
    shape2 = (60, 100)
    s = SkyStream(*shape2, 5, speed=5, streak_period=11, streak_life=7)
    run_pair([s.next() for _ in range(30)], 4, np.ones(shape2, np.uint8), binary_cfg("normal"), pinned=True)


@pytest.mark.parametrize("n", [4, 8, 16, 25, 30])
def test_m3_per_frame_window_kernel_launched_ahead(n, monkeypatch):
    """The window kernel of the NEXT frame is queued by mdh_update behind the current frame's launches and waits on a mailbox in
    pinned host memory (k_stack_threshold_mail); the next mdh_update hands it its arguments instead of launching.  Forced on for
    every frame here (the library only does it for callers in a tight loop): results as on every other path, with get_binary
    between the frames (cancels the waiting kernel: run_pair asks for the pre-mask binary), with a pause longer than the
    kernel's patience (it expires, the frame is launched the usual way), and straight through (the hand-over itself)."""
    import ctypes
    import time
    monkeypatch.setenv("MDH_LEAN_MAIL_GAP", "1e12")
    shape = (72, 192)
    mask = np.ones(shape, np.uint8)
    s = SkyStream(*shape, 77 + n, speed=5, streak_period=11, streak_life=7)
    run_pair([s.next() for _ in range(3 * n + 30)], n, mask, binary_cfg("normal"), pinned=True)
    # no get_binary in between: the waiting kernel takes its frame; every 7th frame comes late
    cfg = binary_cfg("normal")
    s = SkyStream(*shape, 177 + n, speed=5, streak_period=11, streak_life=7)
    frames = np.stack([s.next() for _ in range(3 * n + 40)])
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    pool = det._ctx.host_alloc(frames.nbytes)
    ctypes.memmove(pool, frames.ctypes.data, frames.nbytes)
    pinned = np.ctypeslib.as_array((ctypes.c_uint8 * frames.nbytes).from_address(pool)).reshape(frames.shape)
    got = []
    for t, f in enumerate(pinned):
        if t % 7 == 6:
            time.sleep(0.002)
        det.update(f)
        lines, pred = det.detect()
        got.append((int(det._ctx._info1[0]["bi_threshold"]), int(det._ctx._info1[0]["lit_count"]), det.linesp_ext.reshape(-1, 4).copy(),
                    np.asarray(lines).reshape(-1, 4).copy()))
    total = 0
    for t, f in enumerate(frames):
        orc.update(f)
        o_lines, _ = orc.detect()
        thr, lit, raw, nms = got[t]
        assert thr == orc.bi_threshold and lit == orc.lit_count, f"frame {t}"
        assert np.array_equal(raw, (orc.raw_lines if orc.lines_num <= 500 else orc.raw_lines[:0]).reshape(-1, 4)), f"frame {t}: raw"
        assert np.array_equal(nms, np.asarray(o_lines).reshape(-1, 4)), f"frame {t}: NMS lines"
        total += len(o_lines)
    assert total > 0
    del pinned
    det._ctx.host_free(pool)
    det.close()


def test_m3_per_frame_pinned_interleaved_with_batches():
    """Batched passes leave the slots in the state k_pass_prepare expects, per-frame passes in the clean-after-use state:
    a detector that alternates between the two protocols must keep producing the oracle's results."""
    import ctypes
    shape, n = (128, 192), 6
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg("normal")
    s = SkyStream(*shape, 99, speed=5, streak_period=9, streak_life=6)
    frames = np.stack([s.next() for _ in range(96)])
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    pool = det._ctx.host_alloc(frames.nbytes)
    ctypes.memmove(pool, frames.ctypes.data, frames.nbytes)
    pinned = np.ctypeslib.as_array((ctypes.c_uint8 * frames.nbytes).from_address(pool)).reshape(frames.shape)
    t, total = 0, 0
    plan = [("f", 9), ("b", 20), ("f", 13), ("b", 3), ("f", 25), ("b", 16), ("f", 10)]
    for kind, cnt in plan:
        if kind == "f":
            for k in range(cnt):
                det.update(pinned[t])
                lines, pred = det.detect()
                orc.update(frames[t])
                o_lines, o_pred = orc.detect()
                assert det.bi_threshold == orc.bi_threshold, t
                assert det.lines_num == orc.lines_num, t
                assert np.array_equal(np.asarray(lines).reshape(-1, 4), np.asarray(o_lines).reshape(-1, 4)), t
                assert np.array_equal(det.dst, orc.dst), t
                total += len(o_lines)
                t += 1
        else:
            res, info = det.detect_batch(frames[t:t + cnt])
            for i in range(cnt):
                orc.update(frames[t + i])
                o_lines, o_pred = orc.detect()
                assert int(info[i]["bi_threshold"]) == orc.bi_threshold, t + i
                assert int(info[i]["n_raw"]) == orc.lines_num, t + i
                assert np.array_equal(np.asarray(res[i][0]).reshape(-1, 4), np.asarray(o_lines).reshape(-1, 4)), t + i
                total += len(o_lines)
            t += cnt
    assert total > 0
    del pinned
    det._ctx.host_free(pool)
    det.close()


@pytest.mark.parametrize("shape,n", [((54, 96), 4), ((67, 131), 8), ((64, 64), 3), ((135, 240), 8)])
def test_m3_busy_frames_single(shape, n):
    mask = np.ones(shape, np.uint8)
    # fixed threshold: the busy frames would drive the adaptive one out of range (bright structure inside the noise ROI)
    cfg = binary_cfg(adaptive=False, init_value=9)
    total = run_pair(busy_frames(7 + n, 3 * n + 6, shape), n, mask, cfg)
    assert total > 0


@pytest.mark.parametrize("sens", ["low", "normal", "high"])
def test_m3_sensitivities_and_fixed_threshold(sens):
    shape, n = (108, 192), 5
    mask = np.ones(shape, np.uint8)
    s = SkyStream(*shape, 77, speed=6, streak_period=14, streak_life=8)
    total = run_pair([s.next() for _ in range(45)], n, mask, binary_cfg(sens))
    assert total > 0
    run_pair(busy_frames(22, 2 * n, shape), n, mask, binary_cfg(sens, adaptive=False, init_value=6, dy_mask=False))


def test_m3_threshold_series_matches_reference_golden(golden_dir):
    """The GPU threshold/snr series against the series the REFERENCE produced (tests/golden/g2)."""
    import json, os
    for c in json.load(open(os.path.join(golden_dir, "g2_threshold_series.json"))):
        shape = tuple(c["shape"])
        mask = np.ones(shape, np.uint8)
        cfg = binary_cfg(c["sensitivity"], adaptive=c.get("adaptive", True), init_value=c.get("init_value", 7), area=c["area"],
                         interval=c["interval"])
        det = M3Detector(1, c["n"], mask, 10, cfg, None)
        assert list(det.stack.std_roi) == c["std_roi"]
        frames = np.stack(frames_normal(c["seed"], c["steps"], shape, c["mu"], c["sigma"]))
        _, info = det.detect_batch(frames)
        assert info["bi_threshold"].tolist() == c["bi_threshold"]
        assert np.allclose(info["snr"], c["snr"], rtol=1e-12, atol=0)
        assert np.allclose(info["bi_threshold_float"], c["bi_threshold_float"], rtol=1e-12, atol=0)
        det.close()


def test_m3_sky_stream_batched_equals_oracle():
    H, W, n = 270, 480, 8
    s = SkyStream(H, W, 20220413, speed=10, streak_period=30)
    mask = np.ones((H, W), np.uint8)
    frames = [s.next() for _ in range(70)]
    total = run_pair(frames, n, mask, binary_cfg(), batch=33, max_batch=16)
    assert total > 0


def test_m3_batch_equals_single_frame_protocol():
    shape, n = (90, 160), 6
    mask = np.ones(shape, np.uint8)
    frames = busy_frames(5, 40, shape)
    cfg = binary_cfg(adaptive=False, init_value=9)
    a = M3Detector(1, n, mask, 10, cfg, None)
    b = M3Detector(1, n, mask, 10, cfg, None, max_batch=7)
    singles = []
    for f in frames:
        a.update(f)
        singles.append(a.detect())
    res, _ = b.detect_batch(np.stack(frames))
    for (l1, p1), (l2, p2) in zip(singles, res):
        assert np.array_equal(np.asarray(l1).reshape(-1, 4), np.asarray(l2).reshape(-1, 4)) and np.array_equal(p1, p2)
    a.close(); b.close()


def test_m3_masked_and_ragged_width():
    # width not a multiple of 64, mask with a dark bottom (moves the noise ROI), frames pre-masked like the loader does
    shape, n = (61, 203), 4
    mask = np.ones(shape, np.uint8)
    mask[40:, :] = 0
    frames = [f * mask for f in busy_frames(9, 20, shape)]
    run_pair(frames, n, mask, binary_cfg(area=0.25, adaptive=False, init_value=8))
    s = SkyStream(*shape, 5, speed=5, streak_period=10, streak_life=6, mask01=mask)
    run_pair([s.next() for _ in range(30)], n, mask, binary_cfg(area=0.25))


@pytest.mark.parametrize("shape", [(70, 200), (40, 64 * 65 + 17), (33, 64 * 64), (50, 64 * 128 + 1)])
def test_m3_dense_binaries_direct_mapped_morphology(shape):
    """A lowered fixed threshold lets the noise through: every block of a frame is listed, and the batched path hands such frames
    to k_morph_close_dense (64 adjacent word columns per wave, neighbours' bits through DPP wave shifts, 16-row bands).  Widths
    below, at and across the 64-column wave boundary, ragged last words; passes that mix dense frames with empty ones."""
    n = 4
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=3, hough=(10, 10, 2))
    noise = frames_normal(sum(shape), 14, shape, 30, 2.5)
    flat = [np.full(shape, 30, np.uint8)]
    frames = noise[:6] + flat * 7 + noise[6:10] + flat * 5 + noise[10:]
    run_pair(frames, n, mask, cfg, batch=13, max_batch=13)


@pytest.mark.parametrize("cand", ["1", "2", "0"])
@pytest.mark.parametrize("shape,n", [((70, 192), 4), ((40, 64 * 62), 8), ((33, 64 * 63), 5), ((50, 64 * 125), 16), ((9, 64), 2), ((40, 64 * 33), 25), ((21, 64 * 63), 30)])
def test_m3_dense_output_format_forced(shape, n, cand, monkeypatch):
    """MDH_DENSE=1: every batched pass runs in the dense format (the window kernel writes every word of b plainly, the 64-bit
    direct-mapped morphology kernel k_morph_close_plain reads it: 62 word columns + 2 halo lanes per wave, edge bits through DPP
    wave shifts at every stage) -- noise frames, flat frames, streak frames; widths of one word, below / at / across the 62-column
    wave boundary and across two of them; every window length class of the wide-load kernels."""
    monkeypatch.setenv("MDH_DENSE", "1")
    monkeypatch.setenv("MDH_DENSE_CAND", cand)  # 1 (default): candidate blocks flagged by the window kernel; 2: by a sweep over the plain words; 0: none (round 4)
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=3, hough=(10, 10, 2))
    noise = frames_normal(sum(shape) + n, 12, shape, 30, 2.5)
    flat = [np.full(shape, 30, np.uint8)]
    run_pair(noise[:5] + flat * 4 + noise[5:9] + flat * 3 + noise[9:], n, mask, cfg, batch=11, max_batch=11)
    s = SkyStream(*shape, 77 + n, speed=5, streak_period=9, streak_life=6)
    total = run_pair([s.next() for _ in range(3 * n + 30)], n, mask, binary_cfg("high"), batch=16, max_batch=16)
    assert total > 0 or shape[0] < 20
    # the forced format really ran (and the per-frame protocol stays sparse)
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    det.detect_batch(np.stack(noise[:8]))
    assert det._ctx.dense_passes == 1
    det.update(noise[0])
    det.detect()
    assert det._ctx.dense_passes == 1
    det.close()


@pytest.mark.parametrize("cand", ["1", "2"])
@pytest.mark.parametrize("n", [2, 8])
def test_m3_dense_candidate_blocks_crafted(n, cand, monkeypatch):
    """The dense format's candidate test (round 5; cand 1: inside the window kernel, 2: the sweep kernel): a block is flagged only when one of its words holds two set bits
    at most two columns apart IN ONE ROW -- the least a 3x3 window needs to collect its 5 of 9 -- and the morphology looks at the
    flagged blocks' neighbourhoods only.  Crafted clusters that just make the median, placed where the test has to reach across:
    a word boundary (x = 63 | 64), the boundary between two 16-lane DPP rows (x = 255 | 256: four words per row), a wave boundary
    (x = 1023 | 1024: sixteen words per wave), the image's left / right / top / bottom borders (replicated: a single bit counts
    twice there), clusters whose pair and whose single bits sit in different words and different rows, and clusters one bit short
    (nothing may come out, nothing may be missed next to them).  Isolated noise bits all around keep 9 % of the words busy."""
    monkeypatch.setenv("MDH_DENSE", "1")
    monkeypatch.setenv("MDH_DENSE_CAND", cand)
    H, W = 48, 64 * 36
    mask = np.ones((H, W), np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=9, hough=(10, 10, 2))
    rng = np.random.default_rng(77 + n)
    base = np.full((H, W), 30, np.uint8)

    def lit(points):
        f = base.copy()
        for (y, x) in points:
            if 0 <= y < H and 0 <= x < W:
                f[y, x] = 90
        # isolated bits: one per ~11 words, never within 3 columns / 2 rows of another or of a crafted point
        busy = np.zeros((H, W), bool)
        for (y, x) in points:
            busy[max(0, y - 3):y + 4, max(0, x - 4):x + 5] = True
        for _ in range(H * W // 64 // 11):
            y, x = int(rng.integers(0, H)), int(rng.integers(0, W))
            if not busy[max(0, y - 2):y + 3, max(0, x - 3):x + 4].any():
                f[y, x] = 90
                busy[y, x] = True
        return f

    def cluster(y, x, kind):
        """3x3 window centred at (y, x) with exactly 5 (or 4: "short") samples set, in shapes that split across rows and columns."""
        shapes = {"row3+2": [(0, -1), (0, 0), (0, 1), (1, -1), (1, 0)], "2+2+1": [(-1, -1), (-1, 0), (0, 0), (0, 1), (1, 1)],
                  "col": [(-1, 0), (0, 0), (1, 0), (0, 1), (1, 1)], "2+1+2": [(-1, 0), (-1, 1), (0, -1), (1, 0), (1, 1)],
                  "short": [(-1, -1), (-1, 0), (0, 1), (1, 1)], "pair_far": [(0, -1), (0, 1), (-1, 0), (1, 0), (1, 1)]}
        return [(y + dy, x + dx) for dy, dx in shapes[kind]]

    frames = []
    kinds = ["row3+2", "2+2+1", "col", "2+1+2", "short", "pair_far"]
    xs = [63, 64, 65, 255, 256, 257, 1023, 1024, 1025, 0, 1, W - 1, W - 2, 700, 2047, 2048]
    for k, kind in enumerate(kinds):
        pts = []
        for j, x in enumerate(xs):
            y = [0, 1, H - 1, H - 2, 7, 8, 9, 23, 24][(j + k) % 9]
            pts += cluster(y, x, kind)
        frames.append(lit(pts))
        frames.append(base.copy())
    # border singles: one bit on the replicated left / right / top / bottom border counts twice (2 + 2 + 1 from three bits, etc.)
    frames.append(lit([(10, 0), (11, 0), (12, 0), (20, W - 1), (21, W - 1), (22, W - 1), (0, 300), (0, 301), (1, 300), (H - 1, 900), (H - 1, 901), (H - 2, 901)]))
    frames += [base.copy()] * (n + 1)
    total = run_pair(frames, n, mask, cfg, batch=7, max_batch=7)
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    info, _raw, _nms, _prob = det._ctx.process_batch(np.stack(frames[:8]))
    assert det._ctx.dense_passes == 1 and int(info["lit_count"].sum()) > 0
    det.close()
    assert total >= 0


def test_m3_output_format_flips_mid_stream():
    """The format is chosen per pass from what the last collected pass reported (a sparse pass: half of its frames dense -> dense
    format; a dense pass: fewer than 1.5 % of the words with a set bit -> back): a stream that is quiet, then noisy (the threshold
    EMA is slow, so a jump of the noise level goes straight into the binaries), then a sky with streaks, then quiet again must
    flip to the dense format and back, with the oracle's results on every frame -- image borders included (round 3: the first
    version of the dense morphology kernel lost bottom-row pixels, found by this test) -- and slots that held a dense pass are
    reused by sparse ones and the other way round."""
    shape, n = (72, 256), 4
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg("normal")
    quiet1 = frames_normal(1, 40, shape, 30, 2.0)
    noisy = frames_normal(2, 56, shape, 30, 9.0)
    s = SkyStream(*shape, 3, speed=5, streak_period=9, streak_life=6)
    sky = [s.next() for _ in range(40)]  # (on a frame this small a streak alone keeps 1.5 % of the words busy: still dense)
    quiet2 = frames_normal(4, 40, shape, 30, 2.0)
    frames = np.stack(quiet1 + noisy + sky + quiet2)
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    dense_seen, modes = 0, []
    for t0 in range(0, len(frames), 8):
        before = det._ctx.dense_passes
        res, info = det.detect_batch(frames[t0:t0 + 8])
        modes.append(det._ctx.dense_passes - before)
        for i in range(len(info)):
            orc.update(frames[t0 + i])
            o_lines, o_pred = orc.detect()
            assert int(info[i]["bi_threshold"]) == orc.bi_threshold and int(info[i]["lit_count"]) == orc.lit_count, t0 + i
            assert int(info[i]["n_raw"]) == orc.lines_num, t0 + i
            assert np.array_equal(np.asarray(res[i][0]).reshape(-1, 4), np.asarray(o_lines).reshape(-1, 4)), t0 + i
        assert np.array_equal(det.dst, orc.dst), t0
    det.close()
    # sparse at the start, dense somewhere in the noisy stretch, sparse again at the end
    assert modes[0] == 0 and modes[-1] == 0 and max(modes) == 1, modes
    assert 2 <= sum(modes) <= len(modes) - 6, modes


def test_m3_edge_inputs():
    shape, n = (48, 80), 4
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg()
    frames = [np.zeros(shape, np.uint8)] * 3 + [np.full(shape, 255, np.uint8)] * 3 + busy_frames(3, 6, shape)
    run_pair(frames, n, mask, cfg)
    # n = 1: the window is the frame itself, diff is always 0
    run_pair(busy_frames(4, 5, shape), 1, mask, cfg)


def test_m3_too_many_lines_guard():
    """A frame with > 500 segments is discarded (Detector.py:358-360) but lines_num still reports the count."""
    shape, n = (256, 256), 2
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=5, dy_mask=False, hough=(10, 10, 0))
    base = np.full(shape, 10, np.uint8)
    comb = base.copy()
    for y in range(2, 250, 4):
        for x0 in range(2, 240, 16):
            comb[y:y + 2, x0:x0 + 14] = 250   # ~900 separate 2x14 dashes -> > 700 segments
    det = M3Detector(1, n, mask, 10, cfg, None)
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    for f in (base, comb):
        det.update(f); lines, pred = det.detect()
        orc.update(f); o_lines, o_pred = orc.detect()
    assert orc.lines_num > 500 and det.lines_num == orc.lines_num
    assert len(lines) == 0 and pred.shape == (0, 10) and len(o_lines) == 0
    assert np.array_equal(det.dst, orc.dst)
    det.close()


def test_hough_many_points_global_list():
    """More lit pixels than the LDS point list holds (8192): the global-memory list path."""
    shape, n = (200, 320), 2
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=5, dy_mask=False)
    rng = np.random.default_rng(1)
    base = np.full(shape, 10, np.uint8)
    blob = base.copy()
    blob[20:120, 30:200] = 200          # 17 000 px block
    blob[150:153, 10:300] = 220
    run_pair([base, blob, base], n, mask, cfg)


@pytest.mark.parametrize("n_eff,fps", [(4, 25.0), (3, 49.0)])
@pytest.mark.parametrize("shape", [(80, 144), (72, 192), (40, 64)])
def test_classic_detector(n_eff, fps, shape):
    """Per-frame protocol.  49 fps: int(4 / fps * fps) truncates to 3 and the reference's negative ring indices wrap
    (Detector.py:258-261).  Widths that are a multiple of 64 take the time-blocked wide-load kernel, the others the byte kernel."""
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=12, hough=(10, 10, 3))
    det = ClassicDetector(1, fps, mask, 10, cfg, None)
    assert det.stack_maxsize == int(4 / fps * fps) == n_eff
    orc = mo.ClassicOracle(mask, threshold=12, hough_threshold=10, hough_min_len=10, hough_max_gap=3, hough_fn=co.hough_lines_p,
                           n=n_eff) if n_eff != 4 else \
        mo.ClassicOracle(mask, threshold=12, hough_threshold=10, hough_min_len=10, hough_max_gap=3, hough_fn=co.hough_lines_p)
    got = 0
    for t, f in enumerate(busy_frames(31, 16, shape)):
        det.update(f); lines, pred = det.detect()
        orc.update(f); o_lines, o_pred = orc.detect()
        assert np.array_equal(np.asarray(lines).reshape(-1, 4), o_lines.reshape(-1, 4)), t
        assert np.array_equal(pred, o_pred)
        if t >= 3:
            assert np.array_equal(det.dst, orc.dst), t
        got += len(o_lines)
    assert got > 0 or shape[1] <= 64
    det.close()


@pytest.mark.parametrize("n_eff,fps", [(4, 25.0), (3, 49.0)])
def test_classic_detector_batched_equals_per_frame(n_eff, fps):
    """Passes of many frames (several time chunks of the time-blocked kernel, windows reaching back into the previous pass)
    against the per-frame protocol of a second detector and against the oracle."""
    shape = (96, 256)
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=12, hough=(10, 10, 3))
    frames = np.stack(busy_frames(77, 83, shape))
    a = ClassicDetector(1, fps, mask, 10, cfg, None, max_batch=40)
    b = ClassicDetector(1, fps, mask, 10, cfg, None)
    kw = dict(n=n_eff) if n_eff != 4 else {}
    orc = mo.ClassicOracle(mask, threshold=12, hough_threshold=10, hough_min_len=10, hough_max_gap=3, hough_fn=co.hough_lines_p, **kw)
    res = []
    for lo in (0, 40, 57):
        hi = {0: 40, 40: 57, 57: 83}[lo]
        out, info = a.detect_batch(frames[lo:hi])
        res += out
    got = 0
    for t in range(len(frames)):
        b.update(frames[t]); lines, pred = b.detect()
        orc.update(frames[t]); o_lines, _ = orc.detect()
        assert np.array_equal(np.asarray(res[t][0]).reshape(-1, 4), np.asarray(lines).reshape(-1, 4)), t
        assert np.array_equal(np.asarray(lines).reshape(-1, 4), o_lines.reshape(-1, 4)), t
        got += len(o_lines)
    assert got > 0
    a.close()
    b.close()


def test_classic_sparse_format_equals_dense_kernels_under_a_moving_threshold(monkeypatch):
    """ClassicDetector's sparse format (round 6): tagged difference planes + block lists, and for the ring of four the kernel that
    takes the second difference from the first difference's bits of two frames ago.  That shortcut only holds while T(t) = T(t-2):
    an ADAPTIVE threshold on a sky whose noise level steps up and down makes it move, so the re-fetch path runs in the middle of
    passes (and at every time chunk's first two frames).  The oracle restates the fixed-threshold detect() only, so the three
    variants of the library are compared with each other -- the dense kernels (round 5, pinned against the oracle by the tests
    above) are the reference -- frame by frame: threshold, dst, raw segments, NMS lines; batched passes of two sizes and the
    per-frame protocol.  A noisy stretch under a low threshold lists most blocks: the automatic fall-back to the dense kernels
    and the return to the sparse format are crossed on the way."""
    shape = (96, 256)
    mask = np.ones(shape, np.uint8)
    rng = np.random.default_rng(99)
    s = SkyStream(*shape, 5, speed=6, streak_period=9, streak_life=6)
    frames = []
    for t in range(150):
        f = s.next().astype(np.int16)
        sigma = (0, 3, 9, 1, 14)[(t // 23) % 5]
        if sigma:
            f = f + np.rint(rng.normal(0, sigma, shape)).astype(np.int16)
        frames.append(np.clip(f, 0, 255).astype(np.uint8))
    frames = np.stack(frames)

    def run(sparse, fast4, batch):
        monkeypatch.setenv("MDH_CLASSIC_SPARSE", sparse)
        monkeypatch.setenv("MDH_CLASSIC_FAST4", fast4)
        det = ClassicDetector(1, 25.0, mask, 10, binary_cfg("high"), None, max_batch=64)
        out = []
        if batch:
            for lo in range(0, len(frames), batch):
                res, info = det.detect_batch(frames[lo:lo + batch])
                out += [(int(info[k]["bi_threshold"]), int(info[k]["lit_count"]), int(info[k]["n_raw"]), np.asarray(res[k][0]).reshape(-1, 4).copy())
                        for k in range(len(res))]
        else:
            for f in frames:
                det.update(f)
                lines, _ = det.detect()
                i1 = det._ctx._info1[0]
                out.append((int(i1["bi_threshold"]), int(i1["lit_count"]), int(i1["n_raw"]), np.asarray(lines).reshape(-1, 4).copy()))
        det.close()
        return out

    ref = run("0", "0", 37)
    thresholds = {r[0] for r in ref}
    assert len(thresholds) >= 3, thresholds  # the threshold really moved
    assert sum(r[2] for r in ref) > 0
    for sparse, fast4, batch in (("1", "1", 37), ("1", "1", 64), ("1", "0", 37), ("1", "1", 0), ("0", "0", 0)):
        got = run(sparse, fast4, batch)
        for t, (a, b) in enumerate(zip(ref, got)):
            assert a[:3] == b[:3] and np.array_equal(a[3], b[3]), (sparse, fast4, batch, t, a[:3], b[:3])


def test_api_errors():
    shape = (32, 64)
    mask = np.ones(shape, np.uint8)
    det = M3Detector(1, 4, mask, 10, binary_cfg(), None)
    with pytest.raises(RuntimeError):
        det.detect()
    with pytest.raises(ValueError):
        det.update(np.zeros((31, 64), np.uint8))
    det.update(np.zeros(shape, np.uint8))
    from metdetpy_amd._lib import MetDetHipError
    with pytest.raises(MetDetHipError):
        det._ctx.update(np.zeros(shape, np.uint8))  # two updates in a row
    det.detect()
    det.close()
    with pytest.raises(ValueError):
        M3Detector(1, 300, mask, 10, binary_cfg(), None)


@pytest.mark.parametrize("n", [2, 7, 12, 17, 25, 32, 40])
def test_m3_window_sizes_time_blocked_and_generic(n):
    """Every register-window instantiation family (2..32) and the generic n-reads kernel (n > 32), in batched
    passes long enough to be split into several time chunks, with a width that leaves padding lanes."""
    shape = (40, 150)
    mask = np.ones(shape, np.uint8)
    frames = busy_frames(100 + n, 3 * n + 20, shape)
    total = run_pair(frames, n, mask, binary_cfg(adaptive=False, init_value=9), batch=len(frames), max_batch=len(frames))
    assert total > 0
    s = SkyStream(*shape, n, speed=4, streak_period=9, streak_life=6)
    run_pair([s.next() for _ in range(2 * n + 24)], n, mask, binary_cfg(), batch=2 * n + 24, max_batch=2 * n + 24)


def test_m3_generic_kernel_equals_time_blocked(monkeypatch):
    shape, n = (64, 200), 8
    mask = np.ones(shape, np.uint8)
    frames = np.stack(busy_frames(55, 50, shape))
    cfg = binary_cfg(adaptive=False, init_value=9)
    a = M3Detector(1, n, mask, 10, cfg, None)
    ra, ia = a.detect_batch(frames)
    da = a.dst
    a.close()
    monkeypatch.setenv("MDH_FORCE_GENERIC", "1")
    b = M3Detector(1, n, mask, 10, cfg, None)
    rb, ib = b.detect_batch(frames)
    assert np.array_equal(da, b.dst)
    b.close()
    assert np.array_equal(ia, ib)
    for (l1, p1), (l2, p2) in zip(ra, rb):
        assert np.array_equal(np.asarray(l1).reshape(-1, 4), np.asarray(l2).reshape(-1, 4)) and np.array_equal(p1, p2)


def test_hough_serial_replay_equals_grouped(monkeypatch):
    """MDH_HOUGH_SERIAL=1 selects the one-point-at-a-time PPHT replay; both must equal the oracle."""
    monkeypatch.setenv("MDH_HOUGH_SERIAL", "1")
    shape, n = (135, 240), 8
    mask = np.ones(shape, np.uint8)
    total = run_pair(busy_frames(15, 30, shape), n, mask, binary_cfg(adaptive=False, init_value=9), batch=30, max_batch=30)
    assert total > 0


def test_hough_global_accumulator_path(monkeypatch):
    """MDH_HOUGH_GLOBAL_ACC=1 keeps every accumulator cell in global memory (the path taken when the rho windows
    of the lit pixels' bounding box do not fit in LDS); it must equal the oracle as well."""
    monkeypatch.setenv("MDH_HOUGH_GLOBAL_ACC", "1")
    shape, n = (135, 240), 8
    mask = np.ones(shape, np.uint8)
    total = run_pair(busy_frames(16, 24, shape), n, mask, binary_cfg(adaptive=False, init_value=9), batch=24, max_batch=24)
    assert total > 0


def _hough_lds_needs(d):
    """(points, LDS bytes by the corner windows, LDS bytes by the band windows) of a binary frame: hough3_lds_need of
    metdet_kernels.hpp restated (float32 products, round half to even, angle pairs share the larger window)."""
    ys, xs = np.nonzero(d)
    N = len(ys)
    th = np.arange(180, dtype=np.float64) * (np.pi / 180)
    c, s = np.cos(th).astype(np.float32), np.sin(th).astype(np.float32)
    bx0, bx1, by0, by1 = int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max())

    def rb(x, y):
        return np.rint((np.float32(1) * x * c).astype(np.float32) + (np.float32(1) * y * s).astype(np.float32)).astype(int)
    corners = np.stack([rb(bx0, by0), rb(bx1, by0), rb(bx0, by1), rb(bx1, by1)])
    size_c = corners.max(0) - corners.min(0) + 1
    rows = by1 - by0 + 1
    rmin, rmax = np.full(180, 1 << 30), np.full(180, -(1 << 30))
    for b in range((rows + 7) // 8):
        sel = (ys - by0) // 8 == b
        if not sel.any():
            continue
        xl, xh = int(xs[sel].min()), int(xs[sel].max())
        yl, yh = by0 + 8 * b, min(by0 + 8 * b + 7, by1)
        rmin = np.minimum(rmin, rb(np.where(c >= 0, xl, xh), np.where(s >= 0, yl, yh)))
        rmax = np.maximum(rmax, rb(np.where(c >= 0, xh, xl), np.where(s >= 0, yh, yl)))
    size_b = rmax - rmin + 1
    pair = lambda z: 2 * int(np.maximum(z[0::2], z[1::2]).sum())
    uvc = (min(N, max(bx1 - bx0 + 1, rows) + 2) + 3) & ~3
    mdw = rows * ((bx1 >> 6) - (bx0 >> 6) + 1) * 2
    need = lambda cells: 4 * (64 + ((N + 3) & ~3) + 2 * uvc + mdw + (cells + 1) // 2 + 192)
    return N, need(pair(size_c)), need(pair(size_b))


@pytest.mark.parametrize("bands", ["1", "0"])
def test_hough_band_windows_route_oblique_streaks_to_the_small_launch(bands, monkeypatch):
    """Round 6: two oblique streaks whose rho windows, taken from the corners of their bounding box, need more LDS than the small
    Hough launch has (80 KB) while the windows their pixels really span (x-extent per band of 8 rows) fit: the routing sends the
    frame to the small launch (tier 0: two frames per CU) instead of the large one (tier 1: a CU's whole LDS), the kernel derives the
    same band windows from its point list, and the segments are the oracle's either way (MDH_HOUGH_BANDS=0: round 5's routing)."""
    import ctypes as C
    from metdetpy_amd import _lib
    monkeypatch.setenv("MDH_HOUGH_DEBUG", "1")
    monkeypatch.setenv("MDH_HOUGH_BANDS", bands)
    shape, n = (1080, 1920), 2
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=5, dy_mask=False)
    base = np.full(shape, 10, np.uint8)
    imgs = []
    for L, sep, thick, x0 in ((150, 40, 3, 700), (140, 50, 3, 1200), (130, 60, 3, 200)):
        img = base.copy()
        for k in range(L):
            img[300 + k:300 + k + thick, x0 + k:x0 + k + thick] = 220
            img[300 + k:300 + k + thick, x0 + sep + k:x0 + sep + k + thick] = 220
        imgs.append(img)
    frames = np.stack([base, imgs[0], base, imgs[1], base, imgs[2], base, base])
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    want, routed = [], []
    for t, f in enumerate(frames):
        orc.update(f)
        orc.detect()
        want.append((orc.lines_num, orc.raw_lines.reshape(-1, 4).copy(), orc.lit_count))
        if orc.lit_count > 500:
            N, need_corner, need_band = _hough_lds_needs(orc.dst)
            assert need_band + 2048 < 80 * 1024 < need_corner - 2048 and need_corner < 120 * 1024, (t, N, need_corner, need_band)
            routed.append(t)
    assert len(routed) >= 3, routed  # (a streak stays in the window for n frames)
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=len(frames))
    ctx = det._ctx
    K = len(frames)
    info = np.zeros(K, _lib.FRAME_INFO_DTYPE)
    raw = np.zeros((K, _lib.MDH_MAX_LINES, 4), np.int32)
    ctx.process_batch_into(frames.ctypes.data, K, False, info, raw, None, None)
    dbg = np.zeros((2, K, 8), np.int64)
    assert ctx.L.mdh_debug_hough(C.c_void_p(ctx.handle), dbg.ctypes.data_as(C.POINTER(C.c_longlong)), K) == 0
    for t in range(K):
        assert int(info[t]["lit_count"]) == want[t][2] and int(info[t]["n_raw"]) == want[t][0], t
        assert np.array_equal(raw[t][:want[t][0]], want[t][1][:_lib.MDH_MAX_LINES]), f"frame {t}: raw segments"
    assert sum(w[0] for w in want) >= 15
    tiers = [int(dbg[0, t, 1]) & 0xFF for t in routed]
    assert tiers == [0 if bands == "1" else 1] * len(routed), tiers
    det.close()


def test_hough_windows_too_big_for_lds_fall_back():
    """Lit pixels spread over a 1920x1080 frame: the 180 rho windows of their bounding box exceed the LDS budget."""
    shape, n = (1080, 1920), 2
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg(adaptive=False, init_value=5, dy_mask=False)
    base = np.full(shape, 10, np.uint8)
    img = base.copy()
    img[100:103, 50:400] = 200
    img[900:903, 1500:1900] = 200
    for k in range(300):
        img[300 + k:303 + k, 700 + k:703 + k] = 220
    run_pair([base, img, base], n, mask, cfg, check_act=False)


@pytest.mark.parametrize("n", [2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 15, 16, 17, 20, 24, 25, 29, 30, 31, 32])
def test_m3_wide_load_window_kernels(n):
    """Widths that are multiples of 64 take the wide-load window kernels (16 px/lane for n <= 16, 8 px/lane for
    n = 17 .. 32: the reference's default window_sec 1 is n = 25 / 30); height chosen so the last wave is partly out of range."""
    shape = (37, 192)
    mask = np.ones(shape, np.uint8)
    frames = busy_frames(300 + n, 3 * n + 20, shape)
    total = run_pair(frames, n, mask, binary_cfg(adaptive=False, init_value=9), batch=len(frames), max_batch=len(frames))
    assert total > 0
    s = SkyStream(*shape, n, speed=4, streak_period=9, streak_life=6)
    run_pair([s.next() for _ in range(2 * n + 24)], n, mask, binary_cfg(), batch=17, max_batch=17)


def test_m3_byte_kernel_equals_wide(monkeypatch):
    shape, n = (50, 256), 8
    mask = np.ones(shape, np.uint8)
    frames = np.stack(busy_frames(56, 50, shape))
    cfg = binary_cfg(adaptive=False, init_value=9)
    a = M3Detector(1, n, mask, 10, cfg, None)
    ra, ia = a.detect_batch(frames)
    a.close()
    monkeypatch.setenv("MDH_FORCE_BYTES", "1")
    b = M3Detector(1, n, mask, 10, cfg, None)
    rb, ib = b.detect_batch(frames)
    b.close()
    assert np.array_equal(ia, ib)
    for (l1, p1), (l2, p2) in zip(ra, rb):
        assert np.array_equal(np.asarray(l1).reshape(-1, 4), np.asarray(l2).reshape(-1, 4)) and np.array_equal(p1, p2)


def test_submit_collect_pipeline_equals_process_batch():
    """Two passes in flight (pixel stages of pass p+1 overlap the Hough stage of pass p): same results, same order."""
    shape, n = (90, 192), 6
    mask = np.ones(shape, np.uint8)
    frames = np.stack(busy_frames(77, 60, shape))
    cfg = binary_cfg(adaptive=False, init_value=9)
    a = M3Detector(1, n, mask, 10, cfg, None, max_batch=16)
    want, winfo = a.detect_batch(frames)          # internally chunks of 16, pipelined
    a.close()
    b = M3Detector(1, n, mask, 10, cfg, None, max_batch=16)
    ctx = b._ctx
    from metdetpy_amd import _lib
    got = []
    chunks = [frames[i:i + 16] for i in range(0, 60, 16)]
    bufs = [np.ascontiguousarray(c) for c in chunks]
    out = lambda k: (np.zeros(k, _lib.FRAME_INFO_DTYPE), np.zeros((k, 512, 4), np.int32), np.zeros((k, 512, 4), np.int32),
                     np.zeros((k, 512), np.float64))
    pending = []
    for buf in bufs:
        if ctx.in_flight == ctx.max_in_flight:
            o = out(len(pending[0])); ctx.collect_into(*o); got.append(o); pending.pop(0)
        ctx.submit(buf.ctypes.data, len(buf), False)
        pending.append(buf)
    while pending:
        o = out(len(pending[0])); ctx.collect_into(*o); got.append(o); pending.pop(0)
    info = np.concatenate([g[0] for g in got])
    assert np.array_equal(info, winfo)
    k = 0
    for g in got:
        for i in range(len(g[0])):
            nn = int(g[0][i]["n_nms"])
            assert np.array_equal(g[2][i][:nn], np.asarray(want[k][0]).reshape(-1, 4))
            k += 1
    from metdetpy_amd._lib import MetDetHipError
    with pytest.raises(MetDetHipError):
        ctx.collect_into(*out(1))                 # nothing in flight
    b.close()


def test_hough_one_wave_kernel_still_exact(monkeypatch):
    """MDH_HOUGH_ONE_WAVE=1 selects the one-wave-per-frame kernel (the default is a 256-thread workgroup)."""
    monkeypatch.setenv("MDH_HOUGH_ONE_WAVE", "1")
    shape, n = (135, 240), 8
    mask = np.ones(shape, np.uint8)
    total = run_pair(busy_frames(17, 24, shape), n, mask, binary_cfg(adaptive=False, init_value=9), batch=24, max_batch=24)
    assert total > 0


def test_m3_long_stream_mixed_protocols():
    """1500 frames through one detector, mixing the per-frame protocol, small and large batches and the
    submit/collect pipeline: ring wrap-around, sigma schedule (every interval*n frames), EMA warm-up end and the
    binary ring all have to stay in step with the oracle."""
    H, W, n = 96, 192, 8
    mask = np.ones((H, W), np.uint8)
    cfg = binary_cfg()
    s = SkyStream(H, W, 99, speed=3, streak_period=40, streak_life=10)
    frames = np.stack([s.next() for _ in range(1500)])
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=48)
    cc = co.CM3(n, mask, det.stack.std_roi)
    got_lines = []
    infos = []
    pos = 0
    plan = [("single", 5), ("batch", 13), ("single", 3), ("batch", 200), ("batch", 1), ("batch", 48), ("single", 2),
            ("batch", 97), ("batch", 1131)]
    assert sum(k for _, k in plan) == 1500
    for kind, k in plan:
        chunk = frames[pos:pos + k]
        if kind == "single":
            for f in chunk:
                det.update(f)
                det.detect()
                infos.append(det._ctx._info1[0].copy())
                got_lines.append(det.linesp_ext.reshape(-1, 4).copy())
        else:
            info, raw, nms, prob = det._ctx.process_batch(chunk)
            for i in range(k):
                infos.append(info[i].copy())
                nr = int(info[i]["n_raw"])
                got_lines.append(raw[i][:nr].copy() if nr <= 500 else raw[i][:0].copy())
        pos += k
    total = 0
    for t in range(1500):
        cc.update(frames[t])
        o_raw = cc.detect()
        thr, thr_f, snr, _ = cc.state()
        assert int(infos[t]["bi_threshold"]) == thr, t
        assert float(infos[t]["snr"]) == pytest.approx(snr, rel=1e-12), t
        assert int(infos[t]["lit_count"]) == cc.lit_count, t
        assert np.array_equal(got_lines[t], o_raw.reshape(-1, 4)), t
        total += len(o_raw)
    assert total > 50
    det.close()
