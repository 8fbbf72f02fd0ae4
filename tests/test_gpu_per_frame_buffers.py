"""The per-frame protocol's hand-over of HOST buffers (run on the MI355X box: pytest -m gpu).

The reference's loop pops plain NumPy arrays from its loaders (MetDetPy.py:192; videoloader.py:569-593), and ProcessVideoLoader
recycles a fixed ring of them (a shared RawArray, videoloader.py:696-701, 746-772).  mdh_update reads a frame in place when a
kernel can reach it: a pool of mdh_host_alloc, a range registered with mdh_host_register, or a pageable buffer whose address
came by three times once the loader has opted in (mdh_set_auto_register: the library then registers it itself).  Every path must
give the oracle's results, bit for bit.
"""
import ctypes
import mmap

import numpy as np
import pytest

from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.synth import SkyStream
from metdetpy_amd.utils import select_std_roi
from oracle import c_oracle as co
from oracle import m3_oracle as mo
from tests.helpers import binary_cfg, oracle_kwargs

pytestmark = pytest.mark.gpu


def oracle_series(frames, n, mask, cfg):
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    out = []
    for f in frames:
        orc.update(f)
        lines, _ = orc.detect()
        out.append((orc.bi_threshold, orc.lit_count, (orc.raw_lines if orc.lines_num <= 500 else orc.raw_lines[:0]).reshape(-1, 4).copy(),
                    np.asarray(lines).reshape(-1, 4).copy(), orc.dst.copy()))
    return out


def feed(det, frame):
    det.update(frame)
    lines, _ = det.detect()
    return (int(det._ctx._info1[0]["bi_threshold"]), int(det._ctx._info1[0]["lit_count"]), det.linesp_ext.reshape(-1, 4).copy(),
            np.asarray(lines).reshape(-1, 4).copy())


def same(got, want, t, det=None):
    assert got[0] == want[0] and got[1] == want[1], f"frame {t}: threshold / lit count {got[:2]} vs {want[:2]}"
    assert np.array_equal(got[2], want[2]), f"frame {t}: raw segments"
    assert np.array_equal(got[3], want[3]), f"frame {t}: NMS segments"
    if det is not None:
        assert np.array_equal(det.dst, want[4]), f"frame {t}: dst"


def stream(shape, n, count, seed):
    s = SkyStream(*shape, seed, speed=5, streak_period=11, streak_life=7)
    return np.stack([s.next() for _ in range(count)])


@pytest.mark.parametrize("shape,n,ring", [((72, 192), 8, 4), ((128, 256), 4, 6), ((96, 128), 16, 3)])
def test_recycled_pageable_ring_gets_registered(shape, n, ring):
    """A loader that refills a fixed ring of ordinary NumPy buffers: the third time a buffer comes by it is registered, from then
    on (beyond the first n frames and outside sigma-due frames' general path) its frames are read in place."""
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg("normal")
    frames = stream(shape, n, 5 * n + 40, 11 + n)
    want = oracle_series(frames, n, mask, cfg)
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    det._ctx.set_auto_register(True)  # (the loader says it recycles its buffers and will unregister before freeing one)
    bufs = [np.empty(shape, np.uint8) for _ in range(ring)]
    total = 0
    for t, f in enumerate(frames):
        b = bufs[t % ring]
        b[...] = f
        same(feed(det, b), want[t], t, det if t % 5 == 0 else None)
        total += len(want[t][3])
    st = det._ctx.update_stats()
    assert st["auto_registered"] == ring, st
    assert st["direct_frames"] >= len(frames) - 3 * ring - n - 1, st
    assert st["direct_frames"] + st["copied_frames"] == len(frames)
    assert total > 0
    # the caller frees a buffer: unregister first, then the others keep working
    det._ctx.host_unregister(bufs[0])
    assert det._ctx.update_stats()["auto_registered"] == ring - 1
    det.close()


def test_auto_register_is_opt_in_pageable_frames_are_copied():
    """The default: a buffer that comes by again and again is NOT pinned behind the caller's back (it may be a fresh array per frame
    that malloc keeps handing the same address for -- ADVICE round 4); every frame is copied."""
    shape, n = (72, 192), 4
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg("normal")
    frames = stream(shape, n, 40, 5)
    want = oracle_series(frames, n, mask, cfg)
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    buf = np.empty(shape, np.uint8)
    for t, f in enumerate(frames):
        buf[...] = f
        same(feed(det, buf), want[t], t)
    st = det._ctx.update_stats()
    assert st == {"direct_frames": 0, "copied_frames": len(frames), "auto_registered": 0}
    det.close()


def test_explicitly_registered_ring_and_fresh_arrays():
    """mdh_host_register up front (a loader that knows its ring): in place from the first eligible frame on.  Then frames in
    arrays seen for the first time (ThreadVideoLoader's hand-over): copied, same results; then the ring again."""
    shape, n = (128, 192), 8
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg("normal")
    frames = stream(shape, n, 120, 23)
    want = oracle_series(frames, n, mask, cfg)
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    ring = np.empty((5,) + shape, np.uint8)
    det._ctx.host_register(ring)
    with pytest.raises(_lib.MetDetHipError, match="overlaps"):
        det._ctx.host_register(ring[1])
    keep = []
    for t, f in enumerate(frames):
        if 50 <= t < 80:
            b = f.copy()  # a fresh array per frame, kept alive so that no address repeats
            keep.append(b)
        else:
            b = ring[t % 5]
            b[...] = f
        same(feed(det, b), want[t], t, det if t % 7 == 0 else None)
    st = det._ctx.update_stats()
    assert st["auto_registered"] == 0
    assert st["copied_frames"] >= 30 and st["direct_frames"] >= 120 - 30 - n - 120 // (2 * n) - 2, st
    det._ctx.host_unregister(ring)
    with pytest.raises(_lib.MetDetHipError, match="not a range registered"):
        det._ctx.host_unregister(ring)
    # unregistered: the same buffers are ordinary memory again
    for t in range(3):
        ring[0][...] = frames[t]
        det.update(ring[0])
        det.detect()
    det.close()


def test_registered_buffer_unmapped_and_mapped_again():
    """A frame buffer the library registered on its own, released by its owner in the documented order (mdh_host_unregister, then
    unmap) and replaced by a new mapping -- very likely at the same address: the new buffer starts over as ordinary memory
    (copied twice, then registered again) and its frames are the new contents'."""
    shape, n = (64, 256), 4
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg("normal")
    frames = stream(shape, n, 60, 41)
    want = oracle_series(frames, n, mask, cfg)
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    det._ctx.set_auto_register(True)
    nbytes = shape[0] * shape[1]
    mm = mmap.mmap(-1, nbytes)
    view = np.frombuffer(mm, np.uint8).reshape(shape)
    for t in range(30):
        view[...] = frames[t]
        same(feed(det, view), want[t], t)
    assert det._ctx.update_stats()["auto_registered"] == 1
    addr = view.__array_interface__["data"][0]
    det._ctx.host_unregister((addr, nbytes))  # the documented order: unregister, then free
    del view
    mm.close()
    mm2 = mmap.mmap(-1, nbytes)
    view2 = np.frombuffer(mm2, np.uint8).reshape(shape)
    for t in range(30, 60):
        view2[...] = frames[t]
        same(feed(det, view2), want[t], t)
    det.close()
    del view2
    mm2.close()


def test_sigma_due_frame_with_injected_sums_takes_the_general_path():
    """After mdh_seek + mdh_inject_sums a frame whose sigma is due, handed over from pinned memory, must get its threshold from
    the INJECTED sums (the window restarted empty: its own are incomplete) -- as the batched path does.  (Round 3's direct path for
    due frames ignored the table: wrong threshold for the rest of the chunk, and the entry was never used up.)"""
    shape, n = (72, 192), 4
    mask = np.ones(shape, np.uint8)
    cfg = binary_cfg("normal")              # interval 2: sigma due at t = 2, 3, 4 and then every 8th frame
    interval = cfg.binary.interval
    frames = stream(shape, n, 48, 77)
    det_a = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    pool = det_a._ctx.host_alloc(frames.nbytes)
    ctypes.memmove(pool, frames.ctypes.data, frames.nbytes)
    pinned = np.ctypeslib.as_array((ctypes.c_uint8 * frames.nbytes).from_address(pool)).reshape(frames.shape)
    series = []
    for f in pinned:
        det_a.update(f)
        det_a.detect()
        series.append(det_a._ctx._info1[0].copy())
    # window B: frames 14.. (t0 = 13), frame 16 is due two frames into an empty window
    t0, due = 13, 2 * interval * n
    assert due == 16
    r0, c0, r1, c1 = select_std_roi(mask, cfg.binary.area)
    win = frames[due - n:due, r0:r1, c0:c1].astype(np.int64)
    m = win.sum(0) // n
    s1, s2 = int((win - m).sum()), int(((win - m) ** 2).sum())
    updates = sum(1 for t in range(1, t0 + 1) if (t > n and t % (interval * n) == 0) or (1 < t <= n))
    a = series[t0 - 1]
    det_b = M3Detector(1, n, mask, 10, cfg, None, max_batch=8)
    ctx = det_b._ctx
    ctx.seek(t0, updates, float(a["snr"]), float(a["sigma"]), float(a["bi_threshold_float"]), int(a["bi_threshold"]))
    ctx.inject_sums(due, s1, s2)
    for t in range(t0, len(frames)):
        det_b.update(pinned[t])
        det_b.detect()
        b = ctx._info1[0]
        if t + 1 >= due:
            for f in ("bi_threshold", "bi_threshold_float", "snr", "sigma"):
                assert b[f] == series[t][f], (t + 1, f, b[f], series[t][f])
        if t + 1 >= t0 + 2 * (n - 1) + 1:
            for f in ("n_raw", "n_nms", "lit_count"):
                assert b[f] == series[t][f], (t + 1, f, b[f], series[t][f])
    # the entry has been used up: the table is empty again (64 further entries fit)
    for k in range(64):
        ctx.inject_sums((k + 10) * interval * n, 1, 2)
    del pinned
    det_a._ctx.host_free(pool)
    det_a.close()
    det_b.close()
