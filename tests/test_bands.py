"""Row-band tiling of one stream (metdetpy_amd/bands.py; SURVEY 8e "spatial row-bands", BASELINE config 5).

CPU part: the tiling logic with an engine made of the oracle's functions, in-process (LocalBands) and across two and eight real
processes (BandedM3Detector; collectives over the product's rendezvous store and over gloo), against the whole-frame oracle --
bit-exact.
GPU part (-m gpu): the HIP band / Hough contexts through the C-ABI, against the whole-frame HIP detector and the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from metdetpy_amd.bands import BandPlan, BandedM3Detector, LocalBands
from metdetpy_amd.synth import SkyStream
from oracle import c_oracle as co
from oracle import m3_oracle as mo
from tests.band_oracle_engine import OracleBandEngine, pack_rows, unpack_rows
from tests.helpers import binary_cfg, busy_frames, oracle_kwargs
from tests.mp import run_ranks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def whole_frame_oracle(frames, n, mask, cfg):
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    out = []
    for f in frames:
        orc.update(f)
        lines, pred = orc.detect()
        out.append(dict(lines=np.asarray(lines).reshape(-1, 4), pred=pred, thr=orc.bi_threshold, lit=orc.lit_count,
                        n_raw=orc.lines_num, dst=orc.dst.copy()))
    return out


def make_case(kind, H, W, n, count, seed):
    """'sky': the bench's kind of stream (stars + streaks) under the adaptive threshold, so the all-reduced noise sums
    decide the result; 'busy': dense moving structure under a fixed threshold (the adaptive one would run away)."""
    if kind == "sky":
        frames = SkyStream(H, W, seed, n_stars=40, speed=5, streak_period=9, streak_life=6).take(count)
        return frames, binary_cfg()
    return np.stack(busy_frames(seed, count, (H, W))), binary_cfg(adaptive=False, init_value=9)


def check_against(results, info, want, first=0):
    for i, (lines, pred) in enumerate(results):
        w = want[first + i]
        assert int(info[i]["bi_threshold"]) == w["thr"], (first + i, int(info[i]["bi_threshold"]), w["thr"])
        assert int(info[i]["lit_count"]) == w["lit"], (first + i, int(info[i]["lit_count"]), w["lit"])
        assert int(info[i]["n_raw"]) == w["n_raw"], first + i
        assert np.array_equal(np.asarray(lines).reshape(-1, 4), w["lines"]), first + i
        assert np.array_equal(pred, w["pred"]), first + i


def test_plan_covers_every_row_once():
    for H, N in [(1080, 8), (4320, 8), (97, 3), (100, 7), (64, 1), (9, 2)]:
        p = BandPlan(H, 160, N, (H // 3, 10, H // 3 + H // 4, 90))
        rows = np.zeros(H, int)
        roi_rows = 0
        for d in p.bands:
            rows[d["own0"]:d["own1"]] += 1
            assert d["sub0"] == max(0, d["own0"] - 4) and d["sub1"] == min(H, d["own1"] + 4)
            assert d["own1"] - d["own0"] <= p.band_rows
            roi_rows += d["roi"][2] - d["roi"][0]
        assert (rows == 1).all() and roi_rows == H // 4
    with pytest.raises(ValueError):
        BandPlan(10, 64, 7, (0, 0, 5, 5))  # ceil(10/7) = 2 rows per band leaves bands 5, 6 empty


def test_pack_unpack_rows():
    rng = np.random.default_rng(1)
    img = ((rng.random((5, 150)) < 0.3) * 255).astype(np.uint8)
    w = pack_rows(img, 3)
    assert w.shape == (5, 3) and (int(w[0, 0]) >> 7) & 1 == (img[0, 7] != 0)
    assert np.array_equal(unpack_rows(w, 150), img)


@pytest.mark.parametrize("kind,H,W,n,nbands,K", [("sky", 96, 160, 4, 2, 8), ("busy", 97, 150, 5, 3, 7), ("sky", 120, 128, 8, 5, 10),
                                                 ("busy", 64, 96, 3, 1, 6), ("sky", 90, 200, 6, 3, 12)])
def test_local_bands_equal_whole_frame_oracle(kind, H, W, n, nbands, K):
    mask = np.ones((H, W), np.uint8)
    mask[:H // 6, :W // 3] = 0  # the noise ROI search sees a real mask
    frames, cfg = make_case(kind, H, W, n, 3 * K, H + n)
    frames = frames * mask
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert sum(w["n_raw"] for w in want) > 0
    lb = LocalBands(n / 10, 10, mask, 10, cfg, nbands, max_batch=K, engine=OracleBandEngine)
    for p0 in range(0, len(frames), K):
        res, info = lb.detect_batch(frames[p0:p0 + K])
        check_against(res, info, want, p0)


BAND_WORKER = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, sys.argv[1])
from metdetpy_amd.bands import BandedM3Detector
from tests.band_oracle_engine import OracleBandEngine
from tests.gloo_group import make_group
from tests.test_bands import make_case
use_hip, kind = sys.argv[2] == "hip", sys.argv[3]
H, W, n, K = [int(v) for v in sys.argv[4:8]]
group = make_group(kind)
rank, world = group.rank, group.world
mask = np.ones((H, W), np.uint8)
frames, cfg = make_case("sky", H, W, n, 3 * K, 77)
kw = {} if use_hip else dict(engine=OracleBandEngine)
det = BandedM3Detector(n / 10, 10, mask, 10, cfg, None, rank=rank, world=world, group=group, device=0, max_batch=K, **kw)
out = []
for p0 in range(0, len(frames), K):
    res, info = det.detect_batch(det.plan.crop(frames[p0:p0 + K], rank))
    if rank == 0:
        for (lines, pred), fi in zip(res, info):
            out.append(dict(lines=np.asarray(lines).reshape(-1, 4).tolist(), pred=np.asarray(pred).tolist(),
                            thr=int(fi["bi_threshold"]), lit=int(fi["lit_count"]), n_raw=int(fi["n_raw"])))
    else:
        assert res is None
det.close()
if rank == 0:
    print(json.dumps(out))
group.close()
"""


def run_band_ranks(tmp_path, mode, kind, world, H, W, n, K):
    return run_ranks(tmp_path, "band_worker", BAND_WORKER, world, args=(mode, kind, H, W, n, K))


def compare_worker_output(got, H, W, n, K):
    mask = np.ones((H, W), np.uint8)
    frames, cfg = make_case("sky", H, W, n, 3 * K, 77)
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert len(got) == len(want) and sum(w["n_raw"] for w in want) > 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert (g["thr"], g["lit"], g["n_raw"]) == (w["thr"], w["lit"], w["n_raw"]), i
        assert np.array_equal(np.array(g["lines"], np.int32).reshape(-1, 4), w["lines"]), i
        assert np.array_equal(np.array(g["pred"]).reshape(-1, 10), w["pred"]), i


@pytest.mark.parametrize("kind", ["store", "gloo"])
def test_two_process_bands_equal_whole_frame_oracle(tmp_path, kind):
    """world_size 2: all-reduce of the noise sums, all-to-all of the bit planes, gather of the results -- over the product's own
    group (rendezvous store) and over torch's gloo backend (tests/gloo_group.py)."""
    compare_worker_output(run_band_ranks(tmp_path, "oracle", kind, 2, 135, 192, 6, 9), 135, 192, 6, 9)


@pytest.mark.parametrize("kind", ["store", "gloo"])
def test_eight_process_bands_of_an_8k_shaped_plan(tmp_path, kind):
    """world_size 8, the structure BASELINE config 5 runs with (7680x4320 over 8 GPUs: 540-row bands + 4-row halos, frames r, r + 8,
    ... replayed on rank r) at 1/16 of the linear size: 8 bands of 34 rows (the last one 32) + halos, 12 frames per pass -- so some
    ranks replay two frames of a pass and some one -- three passes, against the whole-frame oracle."""
    H, W, n, K = 270, 480, 6, 12
    from metdetpy_amd.bands import BandPlan
    plan = BandPlan(H, W, 8, (100, 160, 180, 320))
    assert plan.band_rows == 34 and plan.bands[7]["own1"] - plan.bands[7]["own0"] == 32 and plan.bands[3]["sub0"] == 3 * 34 - 4
    compare_worker_output(run_band_ranks(tmp_path, "oracle", kind, 8, H, W, n, K), H, W, n, K)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kind,H,W,n,nbands,K", [("sky", 270, 480, 8, 2, 16), ("busy", 271, 450, 5, 3, 11), ("sky", 540, 960, 8, 8, 24),
                                                 ("busy", 128, 192, 3, 1, 6), ("sky", 96, 128, 16, 4, 8), ("busy", 135, 240, 8, 5, 20)])
def test_hip_bands_equal_whole_frame(kind, H, W, n, nbands, K):
    from metdetpy_amd import M3Detector
    mask = np.ones((H, W), np.uint8)
    mask[:H // 6, :W // 3] = 0
    frames, cfg = make_case(kind, H, W, n, 3 * K, H + n)
    frames = frames * mask
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert sum(w["n_raw"] for w in want) > 0
    lb = LocalBands(n / 10, 10, mask, 10, cfg, nbands, max_batch=K)
    whole = M3Detector(n / 10, 10, mask, 10, cfg, None, max_batch=K)
    for p0 in range(0, len(frames), K):
        res, info = lb.detect_batch(frames[p0:p0 + K])
        check_against(res, info, want, p0)
        wres, winfo = whole.detect_batch(frames[p0:p0 + K])
        for i in range(len(res)):
            assert np.array_equal(np.asarray(res[i][0]).reshape(-1, 4), np.asarray(wres[i][0]).reshape(-1, 4))
            for f in ("n_raw", "n_nms", "lit_count", "bi_threshold", "bi_threshold_float", "snr", "sigma", "dst_sum", "hough_gap"):
                assert info[i][f] == winfo[i][f], (p0 + i, f, info[i][f], winfo[i][f])
        # the owned rows of every band's dst are the whole frame's dst
        d_whole = want[p0 + len(res) - 1]["dst"]
        for b, e in enumerate(lb.engines):
            d = lb.s.plan.bands[b]
            sub = e.band.get_binary(-1, 0)
            assert np.array_equal(sub[d["own0"] - d["sub0"]:d["own1"] - d["sub0"]], d_whole[d["own0"]:d["own1"]]), b
    lb.close()
    whole.close()


@pytest.mark.gpu
def test_hip_synthetic_sky_in_bands():
    """The bench's stream (stars + streaks) in 4 bands, several passes, against the oracle."""
    from tests.helpers import binary_cfg
    H, W, n, K = 272, 480, 8, 32
    cfg = binary_cfg()
    mask = np.ones((H, W), np.uint8)
    frames = SkyStream(H, W, 20220415, speed=8, streak_period=16).take(3 * K)
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert sum(w["n_raw"] for w in want) > 0
    lb = LocalBands(n / 10, 10, mask, 10, cfg, 4, max_batch=K, shared_hough=False)  # (the host path of the all-to-all layout)
    for p0 in range(0, len(frames), K):
        res, info = lb.detect_batch(frames[p0:p0 + K])
        check_against(res, info, want, p0)
    lb.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shared_hough,share_streams", [(True, "1"), (True, "0"), (False, "1")])
def test_hip_bands_resident_and_pipelined(shared_hough, share_streams, monkeypatch):
    """Band rows resident in HBM, two passes in flight, nothing waits for the GPU between submit and collect (queued band steps,
    device-side sum of the noise sums, event-chained hand-over to the Hough stage): same results as the oracle, pass by pass --
    with ONE Hough context fed directly by the bands' export kernels (the single-process default) and with one Hough context
    per band behind an all-to-all of device-to-device copies (the multi-GPU structure).  share_streams: the band contexts of the
    process queue their stages on band 0's streams (mdh_share_streams, the default) or each on its own."""
    from tests.helpers import binary_cfg
    monkeypatch.setenv("MDH_BANDS_SHARE_STREAMS", share_streams)
    H, W, n, K, nbands = 272, 512, 8, 24, 4
    cfg = binary_cfg()
    mask = np.ones((H, W), np.uint8)
    frames = SkyStream(H, W, 20220416, speed=8, streak_period=16).take(5 * K)
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert sum(w["n_raw"] for w in want) > 0
    lb = LocalBands(n / 10, 10, mask, 10, cfg, nbands, max_batch=K, shared_hough=shared_hough)
    plan = lb.s.plan
    passes = []
    for p0 in range(0, len(frames), K):
        ptrs = []
        for b, e in enumerate(lb.engines):
            sub = plan.crop(frames[p0:p0 + K], b)
            ptr = e.band.device_alloc(sub.nbytes)
            e.band.device_upload(ptr, sub)
            ptrs.append(ptr)
        passes.append(ptrs)
    queued = []
    for k, ptrs in enumerate(passes + [None]):
        if ptrs is not None:
            lb.submit_device(ptrs, K)
            queued.append(k)
        if len(queued) == 2 or (ptrs is None and queued):
            first = queued.pop(0) * K
            if shared_hough and first == K:
                # the array form of collect (no per-frame objects): the same frame records, segments in .last_arrays
                res, info = lb.collect(arrays=True)
                assert res is None
                raw, nms, _prob = lb.last_arrays
                for i in range(K):
                    w = want[first + i]
                    assert int(info[i]["n_raw"]) == w["n_raw"] and int(info[i]["bi_threshold"]) == w["thr"], (first + i)
                    assert int(info[i]["lit_count"]) == w["lit"], (first + i)
                    if 0 < w["n_raw"] <= 500:  # (w["lines"]: the oracle's lineset_nms output, as detect() hands it out)
                        assert np.array_equal(nms[i][:int(info[i]["n_nms"])], w["lines"]), (first + i)
                continue
            res, info = lb.collect()
            check_against(res, info, want, first)
    with pytest.raises(IndexError):
        lb.collect()
    for ptrs in passes:
        for b, e in enumerate(lb.engines):
            e.band.device_free(ptrs[b])
    lb.close()


@pytest.mark.gpu
def test_rccl_collectives_on_the_library_streams_one_rank():
    """The multi-GPU data path with its REAL collectives: an RCCL communicator (librccl through ctypes, metdetpy_amd/rccl.py) of
    one rank -- RCCL refuses two ranks on one device, and the box has one -- with the noise all-reduce enqueued in place on the
    band context's noise stream, the bit-plane all-to-all (a send to self) on its post stream, the Hough stage chained behind it by
    events, two passes queued, and no host wait anywhere before collect().  Results against the oracle, pass by pass; the rows
    come from HBM for the odd passes and from host memory for the even ones."""
    from metdetpy_amd import rccl
    from tests.helpers import binary_cfg
    H, W, n, K = 272, 512, 8, 24
    cfg = binary_cfg()
    mask = np.ones((H, W), np.uint8)
    frames = SkyStream(H, W, 20220417, speed=8, streak_period=16).take(4 * K)
    want = whole_frame_oracle(frames, n, mask, cfg)
    assert sum(w["n_raw"] for w in want) > 0
    # timeout_s set: the communicator is made on a helper thread and must still be a BLOCKING one (a non-blocking
    # communicator answers ncclInProgress to the data-path calls below)
    comm = rccl.Communicator(0, 1, rccl.new_unique_id(), device=0, timeout_s=60)
    det = BandedM3Detector(n / 10, 10, mask, 10, cfg, None, rank=0, world=1, max_batch=K, comm=comm)
    band = det.engine.band
    ptrs, queued = [], []
    for k, p0 in enumerate(list(range(0, len(frames), K)) + [None]):
        if p0 is not None:
            sub = det.plan.crop(frames[p0:p0 + K], 0)
            if k % 2:
                ptr = band.device_alloc(sub.nbytes)
                band.device_upload(ptr, sub)
                ptrs.append(ptr)
                det.submit(ptr, K=K, on_device=True)
            else:
                det.submit(sub)
            queued.append(p0)
        if len(queued) == 2 or (p0 is None and queued):
            res, info = det.collect()
            check_against(res, info, want, queued.pop(0))
    for ptr in ptrs:
        band.device_free(ptr)
    det.close()
    comm.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["store", "gloo"])
def test_two_process_hip_bands(tmp_path, kind):
    """Two ranks, each with its own band + Hough contexts on GPU 0 (RCCL refuses two ranks on one device: host-staged collectives
    through the group)."""
    compare_worker_output(run_band_ranks(tmp_path, "hip", kind, 2, 270, 448, 8, 16), 270, 448, 8, 16)


@pytest.mark.gpu
def test_eight_process_hip_bands_on_one_gpu(tmp_path):
    """The 8-rank structure of BASELINE config 5 on the HIP path: eight processes, each with its band + Hough contexts, sharing
    the box's one GPU; 8 bands of a 544-row frame, 24 frames per pass (three replayed per rank), three passes."""
    compare_worker_output(run_band_ranks(tmp_path, "hip", "store", 8, 544, 960, 8, 24), 544, 960, 8, 24)


@pytest.mark.gpu
def test_band_api_errors():
    from metdetpy_amd import M3Detector, _lib
    cfg = binary_cfg()
    mask = np.ones((64, 64), np.uint8)
    det = M3Detector(0.4, 10, mask, 10, cfg, None, max_batch=4)
    with pytest.raises(_lib.MetDetHipError, match="MDH_ROLE_BAND"):
        det._ctx.band_noise(np.zeros((2, 64, 64), np.uint8))
    det.close()
    lb = LocalBands(0.4, 10, mask, 10, cfg, 2, max_batch=4)
    e = lb.engines[0]
    with pytest.raises(_lib.MetDetHipError, match="mdh_band_noise first"):
        e.pixels(np.zeros(2, np.int64))
    with pytest.raises(_lib.MetDetHipError, match="MDH_ROLE_FULL"):
        e.band.process_batch(np.zeros((2, e.band.H, 64), np.uint8))
    # the queued entry points of round 4: thresholds ahead of the pixel stages, export into the shared Hough context
    with pytest.raises(_lib.MetDetHipError, match="mdh_band_submit first"):
        e.band.band_thresholds_submit(lb.s.plan.roi_px)
    with pytest.raises(_lib.MetDetHipError, match="not a MDH_ROLE_HOUGH"):
        e.band.band_pixels_submit_to(lb.s.plan.roi_px, 0, 4, lb.engines[1].band, 0)
    sub = np.zeros((2, e.band.H, 64), np.uint8)
    e.band.band_submit(sub)
    e.band.band_thresholds_submit(lb.s.plan.roi_px)
    with pytest.raises(_lib.MetDetHipError, match="thresholds already"):
        e.band.band_thresholds_submit(lb.s.plan.roi_px)
    with pytest.raises(_lib.MetDetHipError, match="outside the whole frame"):
        e.band.band_pixels_submit_to(lb.s.plan.roi_px, 0, 4, lb.hough, 62)
    e.band.band_skip()
    lb.close()
