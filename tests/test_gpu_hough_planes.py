"""The Hough stage alone on arbitrary binary images: mdh_hough_planes (lit count -> adaptive gap -> HoughLinesP replay ->
lineset_nms) against the oracle's hough_lines_p, far outside what the synthetic sky produces -- lines at every angle, lines
along the image borders, crosses, dense noise, blobs, more than 8192 points (global point list), more than 500 segments,
widths that are not multiples of 64, single rows / columns.  Raw segments must match in discovery order."""
import numpy as np
import pytest

from metdetpy_amd import _lib
from metdetpy_amd.bands import BandPlan, HipBandEngine
from oracle import c_oracle as co
from oracle import m3_oracle as mo
from tests.band_oracle_engine import pack_rows
from tests.helpers import binary_cfg

pytestmark = pytest.mark.gpu


def draw_line(img, x0, y0, x1, y1, width=1):
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    xs = np.rint(np.linspace(x0, x1, n)).astype(int)
    ys = np.rint(np.linspace(y0, y1, n)).astype(int)
    for d in range(width):
        ok = (xs >= 0) & (xs < img.shape[1]) & (ys + d >= 0) & (ys + d < img.shape[0])
        img[ys[ok] + d, xs[ok]] = 255


def family(kind, H, W, rng):
    img = np.zeros((H, W), np.uint8)
    if kind == "fan":  # lines through the centre at many angles
        for a in np.linspace(0, np.pi, 13, endpoint=False):
            r = 0.45 * min(H, W)
            draw_line(img, W / 2 - r * np.cos(a), H / 2 - r * np.sin(a), W / 2 + r * np.cos(a), H / 2 + r * np.sin(a))
    elif kind == "borders":
        img[0, :] = img[-1, :] = 255
        img[:, 0] = img[:, -1] = 255
        draw_line(img, 0, 0, W - 1, H - 1)
    elif kind == "noise":
        img[rng.random((H, W)) < 0.01] = 255
    elif kind == "dense":
        img[rng.random((H, W)) < 0.12] = 255
    elif kind == "segments":
        for _ in range(40):
            x, y = rng.uniform(0, W), rng.uniform(0, H)
            a, ln = rng.uniform(0, np.pi), rng.uniform(5, 60)
            draw_line(img, x, y, x + ln * np.cos(a), y + ln * np.sin(a), width=int(rng.integers(1, 3)))
        img[rng.random((H, W)) < 0.002] = 255
    elif kind == "dashed":  # gaps around the maxLineGap
        for row, gap in ((H // 4, 3), (H // 2, 9), (3 * H // 4, 12)):
            x = 2
            while x + 12 < W:
                img[row, x:x + 12] = 255
                x += 12 + gap
    elif kind == "blobs":
        for _ in range(12):
            x, y, r = int(rng.uniform(0, W)), int(rng.uniform(0, H)), int(rng.integers(2, 9))
            yy, xx = np.ogrid[:H, :W]
            img[(yy - y)**2 + (xx - x)**2 <= r * r] = 255
    elif kind == "grid":  # many short segments: more than 500 lines on big images
        img[::6, :] = 255
        img[:, ::37] = 0
    elif kind == "cloud":  # a cloud edge: a wavy strip across the whole frame (12 k+ points: the one-wave kernel with its list in LDS) + a streak
        xs = np.arange(W)
        ys = np.rint(H * (0.5 + 0.08 * np.sin(xs / W * 2 * np.pi * rng.uniform(1, 3) + rng.uniform(0, 6)))).astype(int)
        for d in range(max(3, -(-11520 // W))):
            img[np.clip(ys + d, 0, H - 1), xs] = 255
        draw_line(img, W * 0.2, H * 0.2, W * 0.2 + 150, H * 0.2 + 90, width=3)
    elif kind == "single":
        img[H // 2, W // 3] = 255
    elif kind == "empty":
        pass
    return img


CASES = [("fan", 200, 320), ("borders", 97, 131), ("noise", 270, 480), ("dense", 120, 200), ("segments", 300, 500), ("dashed", 90, 400),
         ("blobs", 128, 128), ("grid", 360, 640), ("single", 33, 70), ("empty", 64, 64), ("segments", 1, 300), ("segments", 300, 1),
         ("dense", 300, 420), ("fan", 1080, 1920), ("segments", 540, 960), ("cloud", 1080, 1920), ("cloud", 540, 960)]


@pytest.mark.parametrize("mask_share", [1.0, 0.02])
def test_hough_planes_equal_oracle(mask_share):
    """mask_share scales mask_area: 0.02 drives the lit share over 0.05 % (adaptive gap 0); 1.0 keeps gaps up to max_gap."""
    cfg = binary_cfg()
    rng = np.random.default_rng(17)
    seen_tiers = dict(big=0, over500=0, gap0=0, gapmax=0)
    for kind, H, W in CASES:
        imgs = [family(kind, H, W, rng) for _ in range(3)]
        K = len(imgs)
        plan = BandPlan(H, W, 1, (0, 0, H, W))
        mask_area = max(1.0, H * W * mask_share)
        eng = HipBandEngine(plan, 0, 2, cfg, mask_area, K, 0)
        planes = np.stack([pack_rows(im, plan.WW) for im in imgs])[None]  # (1, K, H, WW)
        info, raw, nms, prob = eng.hough_stage(planes, K)
        for j, im in enumerate(imgs):
            lit = int(np.count_nonzero(im))
            dst_sum = float(np.sum(im / 255.) / mask_area * 100)
            gap = max(0, 1 - dst_sum / 0.05) * cfg.hough_line.max_gap
            want = co.hough_lines_p(im, 1, np.pi / 180.0, cfg.hough_line.threshold, cfg.hough_line.min_len, gap)
            assert int(info[j]["lit_count"]) == lit and info[j]["dst_sum"] == dst_sum and info[j]["hough_gap"] == gap, (kind, H, W, j)
            assert int(info[j]["n_raw"]) == len(want), (kind, H, W, j, int(info[j]["n_raw"]), len(want))
            k = min(len(want), _lib.MDH_MAX_LINES)
            assert np.array_equal(raw[j, :k], want[:k]), (kind, H, W, j)
            if 0 < len(want) <= 500:
                kept, nonline = mo.lineset_nms(want)
                assert int(info[j]["n_nms"]) == len(kept) and np.array_equal(nms[j, :len(kept)], kept)
                assert np.array_equal(prob[j, :len(kept)], nonline)
            else:
                assert int(info[j]["n_nms"]) == 0
            seen_tiers["big"] += lit > 8192
            seen_tiers["over500"] += len(want) > 500
            seen_tiers["gap0"] += gap == 0 and lit > 0
            seen_tiers["gapmax"] += gap > 5
        eng.close()
    assert seen_tiers["big"] > 0 and seen_tiers["gap0" if mask_share < 1 else "gapmax"] > 0, seen_tiers


@pytest.mark.parametrize("shape", [(540, 960), (270, 1920)])
def test_hough_point_counts_at_the_kernel_boundaries(shape):
    """Frames with exactly 8191 ... 8193 and 32767 ... 32769 lit pixels: the last frame of the three-wave kernels' LDS point list, the
    first and the last of the hybrid replay with a visiting-order table of the frame's own (HOUGH_BIG_LIST), the first of the one-wave
    kernel -- scattered pixels plus two streaks, so that every route has segments to find, in one pass so that the classifier sends
    neighbours down different routes side by side."""
    H, W = shape
    cfg = binary_cfg()
    rng = np.random.default_rng(8192)
    counts = [8191, 8192, 8193, 32767, 32768, 32769]
    imgs = []
    for N in counts:
        im = np.zeros((H, W), np.uint8)
        draw_line(im, W * 0.1, H * 0.15, W * 0.1 + 180, H * 0.15 + 70, width=3)
        draw_line(im, W * 0.7, H * 0.8, W * 0.7 + 120, H * 0.8 - 95, width=2)
        free = np.flatnonzero(im.ravel() == 0)
        need = N - int(np.count_nonzero(im))
        im.ravel()[rng.choice(free, need, replace=False)] = 255
        assert int(np.count_nonzero(im)) == N
        imgs.append(im)
    K = len(imgs)
    plan = BandPlan(H, W, 1, (0, 0, H, W))
    mask_area = float(H * W)
    eng = HipBandEngine(plan, 0, 2, cfg, mask_area, K, 0)
    planes = np.stack([pack_rows(im, plan.WW) for im in imgs])[None]
    info, raw, nms, prob = eng.hough_stage(planes, K)
    for j, im in enumerate(imgs):
        dst_sum = float(np.sum(im / 255.) / mask_area * 100)
        gap = max(0, 1 - dst_sum / 0.05) * cfg.hough_line.max_gap
        want = co.hough_lines_p(im, 1, np.pi / 180.0, cfg.hough_line.threshold, cfg.hough_line.min_len, gap)
        assert int(info[j]["lit_count"]) == counts[j]
        assert int(info[j]["n_raw"]) == len(want) and len(want) > 0, (counts[j], int(info[j]["n_raw"]), len(want))
        k = min(len(want), _lib.MDH_MAX_LINES)
        assert np.array_equal(raw[j, :k], want[:k]), counts[j]
    eng.close()
