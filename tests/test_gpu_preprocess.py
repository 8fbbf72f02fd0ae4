"""Loader-side preprocessing and clip max-stack on the GPU vs the oracle's restatements (bit-exact).
The OpenCV fixed-point paths themselves are PARITY UNPINNED (no cv2 anywhere), see DESIGN.md section 2."""
import numpy as np
import pytest

from metdetpy_amd.imgproc import Preprocessor, load_mask, max_stack
from oracle import m3_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src_hw,dst_wh,exp", [((216, 384), (96, 54), 1), ((216, 384), (96, 54), 3), ((135, 240), (240, 135), 2),
                                                ((101, 203), (77, 41), 2), ((64, 80), (160, 128), 1), ((540, 960), (480, 270), 4)])
def test_preprocess_matches_oracle(src_hw, dst_wh, exp):
    rng = np.random.default_rng(sum(src_hw) + exp)
    frames = rng.integers(0, 256, (exp, src_hw[0], src_hw[1], 3), dtype=np.uint8)
    mask = (rng.random((dst_wh[1], dst_wh[0])) < 0.9).astype(np.uint8)
    for m in (mask, None):
        pre = Preprocessor(src_hw, dst_wh, m)
        got = pre(frames)
        ones = np.ones((dst_wh[1], dst_wh[0]), np.uint8)
        want = mo.merge_max([mo.preprocess_frame(f, dst_wh, m if m is not None else ones) for f in frames])
        assert np.array_equal(got, want), np.argwhere(got != want)[:5]
        pre.close()


@pytest.mark.parametrize("prefetch", ["1", "0"])
@pytest.mark.parametrize("src_hw,dst_wh", [((72, 1536), (384, 18)),    # 4x: a 3 KB row span per workgroup -- one 16-byte piece per lane
                                           ((40, 3840), (480, 10)),    # 8x: 6 KB spans -- two pieces per lane
                                           ((24, 7680), (512, 6)),     # 15x: 11.5 KB spans -- the kernel without the register prefetch
                                           ((54, 1000), (250, 27))])   # rows that are no multiple of 16 bytes: the one-thread-per-pixel kernel
def test_preprocess_batch_kernels_by_span_and_prefetch(src_hw, dst_wh, prefetch, monkeypatch):
    """Round 6: the batched launch requests the next source frame's rows into registers while the current one is computed
    (k_preprocess_rows_pf, one or two 16-byte pieces per lane and row span; MDH_PRE_PREFETCH=0: the kernel without).  Seven output
    frames of three source frames each -- two workgroup rounds of four and three frames, 21 stages -- per geometry of each kernel."""
    monkeypatch.setenv("MDH_PRE_PREFETCH", prefetch)
    exp, n_out = 3, 7
    rng = np.random.default_rng(src_hw[1] + int(prefetch))
    bgr = rng.integers(0, 256, (exp * n_out, src_hw[0], src_hw[1], 3), dtype=np.uint8)
    mask = (rng.random((dst_wh[1], dst_wh[0])) < 0.9).astype(np.uint8)
    pre = Preprocessor(src_hw, dst_wh, mask)
    got = pre.batch(bgr, exp)
    want = np.stack([mo.merge_max([mo.preprocess_frame(f, dst_wh, mask) for f in bgr[i * exp:(i + 1) * exp]]) for i in range(n_out)])
    assert np.array_equal(got, want), np.argwhere(got != want)[:5]
    pre.close()


def test_preprocess_smooth_image_and_identity():
    y, x = np.mgrid[0:120, 0:200]
    img = np.stack([(x + y) % 256, (2 * x) % 256, (3 * y) % 256], axis=-1).astype(np.uint8)
    pre = Preprocessor((120, 200), None, None)
    assert np.array_equal(pre(img), mo.bgr_to_gray(img))
    pre.close()


@pytest.mark.parametrize("shape,K", [((4, 37, 53, 3), 4), ((9, 64, 64), 9), ((1, 10, 7, 3), 1), ((5, 5), 5)])
def test_max_stack(shape, K):
    rng = np.random.default_rng(K)
    frames = rng.integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(max_stack(frames), mo.max_stack(list(frames)))


def test_load_mask_jpg_and_png():
    rng = np.random.default_rng(4)
    y, x = np.mgrid[0:135, 0:240]
    smooth = (128 + 100 * np.sin(x / 17.0) * np.cos(y / 11.0)).astype(np.uint8)
    bgr = np.stack([smooth, np.roll(smooth, 5, 1), np.roll(smooth, 9, 0)], axis=-1)
    for dsize in (None, (96, 54), (480, 270)):
        assert np.array_equal(load_mask(bgr, "jpg", dsize), mo.load_mask_jpg(bgr, dsize))
    bgra = np.concatenate([bgr, rng.integers(0, 256, (135, 240, 1), dtype=np.uint8)], axis=2)
    bgra[40:90, 60:200, 3] = 255
    for dsize in (None, (120, 68)):
        got = load_mask(bgra, "png", dsize)
        assert np.array_equal(got, mo.load_mask_png(bgra, dsize)) and 0 < got.mean() < 1
    assert load_mask(None, dsize_wh=(32, 20)).shape == (20, 32) and load_mask(None, dsize_wh=(32, 20), grayscale=False).shape == (20, 32, 3)


def test_preprocess_batch_feeds_the_detector_on_device():
    """BGR clip -> resize / gray / mask / exposure merge for a whole pass in one launch, output left in HBM and consumed in
    place by the detector; same detections as the oracle chain on the CPU."""
    from metdetpy_amd import M3Detector
    from metdetpy_amd.synth import SkyStream
    from oracle import c_oracle as co
    from tests.helpers import binary_cfg, oracle_kwargs
    sh, sw, dw, dh, exp, n_out, n = 216, 384, 192, 108, 2, 24, 4
    gray = SkyStream(sh, sw, 11, n_stars=30, speed=9, streak_period=10, streak_life=8).take(n_out * exp)
    rng = np.random.default_rng(2)
    bgr = np.clip(gray[..., None].astype(np.int16) + rng.integers(-3, 4, gray.shape + (3, )), 0, 255).astype(np.uint8)
    mask = np.ones((dh, dw), np.uint8)
    mask[:10] = 0
    pre = Preprocessor((sh, sw), (dw, dh), mask)
    want_frames = np.stack([mo.merge_max([mo.preprocess_frame(f, (dw, dh), mask) for f in bgr[i * exp:(i + 1) * exp]])
                            for i in range(n_out)])
    assert np.array_equal(pre.batch(bgr, exp), want_frames)
    cfg = binary_cfg()
    det = M3Detector(n / 10, 10, mask, 10, cfg, None, max_batch=n_out)
    ctx = det._ctx
    d_src = ctx.device_alloc(bgr.nbytes)
    d_out = ctx.device_alloc(n_out * dh * dw)
    ctx.device_upload(d_src, bgr)
    pre.batch_device(d_src, n_out, exp, d_out)
    res, info = det.detect_batch(d_out, K=n_out, on_device=True)
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    segs = 0
    for i, f in enumerate(want_frames):
        orc.update(f)
        lines, pred = orc.detect()
        assert int(info[i]["bi_threshold"]) == orc.bi_threshold and int(info[i]["lit_count"]) == orc.lit_count
        assert np.array_equal(np.asarray(res[i][0]).reshape(-1, 4), np.asarray(lines).reshape(-1, 4))
        segs += len(lines)
    assert segs > 0
    ctx.device_free(d_src)
    ctx.device_free(d_out)
    det.close()
    pre.close()
