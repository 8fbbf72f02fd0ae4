"""hough_mode = "accumulator" (MDH_HOUGH_ACCUMULATOR): the order-free full-accumulator transform next to the exact replay of
cv2.HoughLinesP.  It is a different algorithm, so nothing here is bit-for-bit; the bar is SURVEY 8d's tolerance rule:

  * isolated lines: the segment it returns lies on the drawn line -- both end points within 1 px of it (rho), theta within 0.5
    degrees of the ground truth, at least half of the drawn extent in the first (strongest) segment -- at every angle, and its segments together span at
    least 90 % of what the exact replay's segments span on the same image;
  * a synthetic sky stream through detector + collector: the same meteors, matched with evaluate.py's rule (time IoU >= 0.3 and
    box IoU >= 0.3), as the exact mode;
  * determinism: the result is a function of the binary image (two runs, and the per-frame protocol vs batched passes, agree).
"""
import math

import numpy as np
import pytest

from metdetpy_amd import M3Detector, _lib
from metdetpy_amd.collector import NativeMeteorCollector
from metdetpy_amd.metstruct import CollectorCfg, MeteorCfg, RecheckSwitch, RuntimeParams
from metdetpy_amd.pipeline import calculate_area_iou, calculate_time_iou
from metdetpy_amd.synth import SkyStream
from tests.band_oracle_engine import pack_rows
from tests.helpers import binary_cfg

pytestmark = pytest.mark.gpu


def hough_context(H, W, K, mode, cfg, mask_area):
    c = _lib.MdhCfg()
    c.detector = _lib.MDH_DET_M3
    c.adaptive_bi_thre = 1
    c.init_value = 7
    c.sensitivity = _lib.SENSITIVITY_ID["normal"]
    c.interval = 2
    c.hough_threshold = int(cfg.hough_line.threshold)
    c.hough_min_len = int(cfg.hough_line.min_len)
    c.hough_max_gap = int(cfg.hough_line.max_gap)
    c.dy_mask = 1
    c.max_batch = K
    c.role = _lib.MDH_ROLE_HOUGH
    c.mask_area = float(mask_area)
    c.hough_mode = mode
    return _lib.Context(H, W, 1, c)


def line_image(H, W, x0, y0, x1, y1):
    img = np.zeros((H, W), np.uint8)
    n = 3 * (int(max(abs(x1 - x0), abs(y1 - y0))) + 1)  # oversampled: no holes along the major axis
    xs = np.rint(np.linspace(x0, x1, n)).astype(int)
    ys = np.rint(np.linspace(y0, y1, n)).astype(int)
    img[ys, xs] = 255
    return img


def rho_theta(x0, y0, x1, y1):
    """Normal form of the line through two points: theta in [0, pi), rho signed (OpenCV's convention)."""
    th = math.atan2(x1 - x0, -(y1 - y0)) % math.pi  # direction (dx, dy) -> normal angle
    return x0 * math.cos(th) + y0 * math.sin(th), th


@pytest.mark.parametrize("mask_share", [1.0, 0.02])
def test_accumulator_segments_equal_the_oracle_bit_for_bit(mask_share):
    """The mode's specification is oracle/hough_accumulator.py (a function of the binary image); the HIP path must return the same
    segments in the same order on the image families of tests/test_gpu_hough_planes.py -- fans, borders, noise, dense noise,
    short segments, dashes around the gap limit, blobs, grids (> 500 segments), single pixels, 1-pixel-wide images, > 8192 lit
    pixels (both modes replay cv2.HoughLinesP there) -- at both ends of the adaptive gap."""
    from oracle import hough_accumulator as ha
    from tests.test_gpu_hough_planes import CASES, family
    cfg = binary_cfg()
    rng = np.random.default_rng(23)
    seen = dict(lines=0, over_list=0, over500=0, multi=0)
    for kind, H, W in CASES:
        imgs = [family(kind, H, W, rng) for _ in range(3)]
        K = len(imgs)
        WW = (W + 63) // 64
        mask_area = max(1.0, H * W * mask_share)
        ctx = hough_context(H, W, K, _lib.MDH_HOUGH_ACCUMULATOR, cfg, mask_area)
        planes = np.stack([pack_rows(im, WW) for im in imgs])[None]
        info, raw = ctx.hough_planes(planes, K, 1, H)[:2]
        ctx.close()
        for j, im in enumerate(imgs):
            gap = ha.adaptive_gap(im, mask_area, cfg.hough_line.max_gap)
            want = ha.c_hough_lines_acc(im, cfg.hough_line.threshold, cfg.hough_line.min_len, gap)
            assert info[j]["hough_gap"] == gap
            assert int(info[j]["n_raw"]) == len(want), (kind, H, W, j, int(info[j]["n_raw"]), len(want))
            k = min(len(want), _lib.MDH_MAX_LINES)
            assert np.array_equal(raw[j, :k], want[:k]), (kind, H, W, j)
            lit = int(np.count_nonzero(im))
            seen["lines"] += len(want)
            seen["over_list"] += lit > ha.MAX_LIST
            seen["over500"] += len(want) > 500
            seen["multi"] += len(want) > 3
    assert seen["lines"] > 100 and seen["over_list"] > 0 and seen["multi"] > 5, seen


def test_accumulator_stream_equals_the_oracle():
    """Detector + accumulator mode on a sky stream: every frame's raw segments equal the oracle's on the frame's own binary image
    (fetched from the GPU), i.e. the mode is pinned inside the whole path too, not only behind mdh_hough_planes."""
    from oracle import hough_accumulator as ha
    H, W, n = 270, 480, 8
    mask = np.ones((H, W), np.uint8)
    frames = SkyStream(H, W, 777, speed=9, streak_period=30, streak_life=12).take(96)
    cfg = binary_cfg("normal")
    det = M3Detector(1, n, mask, 10, cfg, None, hough_mode="accumulator")
    total = 0
    for t in range(len(frames)):
        det.update(frames[t])
        det.detect()
        dst = det.dst
        gap = ha.adaptive_gap(dst, float(mask.sum()), cfg.hough_line.max_gap)
        want = ha.c_hough_lines_acc(dst, cfg.hough_line.threshold, cfg.hough_line.min_len, gap)
        assert det.lines_num == len(want), t
        if 0 < len(want) <= 500:
            assert np.array_equal(np.asarray(det.linesp_ext).reshape(-1, 4), want), t
        total += len(want)
    det.close()
    assert total > 10


def test_isolated_lines_against_ground_truth():
    H, W = 240, 320
    cfg = binary_cfg()
    cases = []
    for deg in range(0, 180, 9):
        a = math.radians(deg + 0.3)
        cx, cy, half = 160 + (deg % 5) * 3, 120 - (deg % 7) * 2, 70
        cases.append((cx - half * math.cos(a), cy - half * math.sin(a), cx + half * math.cos(a), cy + half * math.sin(a)))
    imgs = [line_image(H, W, *c) for c in cases]
    K = len(imgs)
    planes = np.stack([pack_rows(im, (W + 63) // 64) for im in imgs])[None]
    out = {}
    for mode in (_lib.MDH_HOUGH_EXACT, _lib.MDH_HOUGH_ACCUMULATOR):
        ctx = hough_context(H, W, K, mode, cfg, 1e9)  # a huge mask area keeps the adaptive maxLineGap at max_gap
        out[mode] = ctx.hough_planes(planes, K, 1, H)
        ctx.close()
    info_e, raw_e = out[_lib.MDH_HOUGH_EXACT][:2]
    info_a, raw_a = out[_lib.MDH_HOUGH_ACCUMULATOR][:2]
    bad = []  # every violation of the run, not just the first one

    def span(raw, count):
        """extent of the union of a frame's segments along the major axis of the drawn line"""
        pts = raw[:min(count, 512)].astype(float).reshape(-1, 2)
        return float(max(np.ptp(pts[:, 0]), np.ptp(pts[:, 1])))

    for j, (x0, y0, x1, y1) in enumerate(cases):
        na, ne = int(info_a[j]["n_raw"]), int(info_e[j]["n_raw"])
        if na < 1 or ne < 1:
            bad.append((j, "no segment", na, ne))
            continue
        seg = raw_a[j, 0].astype(float)  # the strongest line comes out first
        rho_t, th_t = rho_theta(x0, y0, x1, y1)
        _, th_s = rho_theta(*seg)
        dth = min(abs(th_s - th_t), math.pi - abs(th_s - th_t))
        # (0.5 degrees for the transform + what rounding an end point to whole pixels can turn a segment of this length by)
        if math.degrees(dth) > 0.5 + math.degrees(math.atan(1.0 / max(1.0, math.hypot(seg[2] - seg[0], seg[3] - seg[1])))):
            bad.append((j, "theta", math.degrees(th_t), math.degrees(th_s)))
        # rho within 1 px where the segment is: the distance of both of its end points from the true line (rho taken at the
        # origin would add theta's half-degree times a lever arm of a few hundred pixels)
        for ex, ey in ((seg[0], seg[1]), (seg[2], seg[3])):
            if abs(ex * math.cos(th_t) + ey * math.sin(th_t) - rho_t) > 1.0:
                bad.append((j, "rho", ex, ey, rho_t, math.degrees(th_t)))
        # extent: the walk follows a 1-degree grid angle along a one-pixel digital line, so (like the exact replay) it may give a
        # long line up in two or three overlapping events, and the strongest cell need not be the one with the longest walk: the
        # first segment covers at least half of the drawn line, and all of them together cover what the exact replay's segments cover
        true_len = max(abs(x1 - x0), abs(y1 - y0))
        if max(abs(seg[2] - seg[0]), abs(seg[3] - seg[1])) < 0.5 * true_len:
            bad.append((j, "first segment short", seg.tolist(), (x0, y0, x1, y1)))
        if span(raw_a[j], na) < 0.9 * span(raw_e[j], ne):
            bad.append((j, "union short", raw_a[j, :na].tolist(), raw_e[j, :ne].tolist()))
    assert not bad, bad


def _records(frames, mask, n, cfg, mode):
    H, W = mask.shape
    det = M3Detector(1, n, mask, 10, cfg, None, max_batch=64, hough_mode=mode)
    info, raw, nms, prob = det._ctx.process_batch(frames)
    det.close()
    ccfg = CollectorCfg(MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})
    rt = RuntimeParams(fps=float(n), exp_frame=1, eq_fps=float(n), eq_int_fps=n, exp_time=1.0 / n, runtime_size=[W, H],
                       raw_size=[W, H], positive_category_list=["METEOR", "RED_SPRITE"])
    col = NativeMeteorCollector(ccfg, rt)
    col.update_batch(0, 0, info, nms, prob)
    col.clear()
    return [t for r in col.met_exporter.meteor_list for t in r.target], info, raw


class _Box:
    def __init__(self, t):
        xs, ys = sorted([t.pt1[0], t.pt2[0]]), sorted([t.pt1[1], t.pt2[1]])
        self.x1, self.x2, self.y1, self.y2 = xs[0], xs[1], ys[0], ys[1]


def test_stream_detections_match_exact_mode():
    H, W, n = 270, 480, 8
    mask = np.ones((H, W), np.uint8)
    frames = SkyStream(H, W, 4242, speed=9, streak_period=40, streak_life=12).take(360)
    cfg = binary_cfg("normal")
    rec_e, info_e, _ = _records(frames, mask, n, cfg, "exact")
    rec_a, info_a, raw_a = _records(frames, mask, n, cfg, "accumulator")
    # everything in front of the Hough stage is shared
    for f in ("bi_threshold", "lit_count", "hough_gap"):
        assert np.array_equal(info_e[f], info_a[f]), f
    assert len(rec_e) >= 5 and len(rec_a) == len(rec_e), (len(rec_e), len(rec_a))
    for a in rec_a:
        best = max(rec_e, key=lambda e: calculate_time_iou(a, e))
        assert calculate_time_iou(a, best) >= 0.3 and calculate_area_iou(_Box(a), _Box(best)) >= 0.3, (a.to_dict(), best.to_dict())
    assert int(info_a["n_raw"].sum()) > 0
    # determinism, and the per-frame protocol takes the same path
    rec_b, info_b, raw_b = _records(frames, mask, n, cfg, "accumulator")
    assert np.array_equal(info_a, info_b) and np.array_equal(raw_a, raw_b)
    det = M3Detector(1, n, mask, 10, cfg, None, hough_mode="accumulator")
    for t in range(60):
        det.update(frames[t])
        det.detect()
        k = min(int(info_a[t]["n_raw"]), 512)
        assert det.lines_num == int(info_a[t]["n_raw"]), t
        assert np.array_equal(np.asarray(det.linesp_ext).reshape(-1, 4)[:k], raw_a[t, :k]), t
    det.close()


def test_bad_mode_is_refused():
    with pytest.raises(ValueError):
        M3Detector(1, 4, np.ones((32, 64), np.uint8), 10, binary_cfg(), None, hough_mode="fast")
    c = _lib.MdhCfg()
    c.detector = _lib.MDH_DET_M3
    c.sensitivity, c.interval, c.max_batch, c.mask_area = 1, 2, 4, 32.0 * 64
    c.roi[2], c.roi[3] = 32, 64
    c.hough_threshold = c.hough_min_len = c.hough_max_gap = 10
    c.hough_mode = 7
    with pytest.raises(_lib.MetDetHipError, match="hough_mode"):
        _lib.Context(32, 64, 4, c)
