"""Loader-side preprocessing and clip max-stack on the GPU vs the oracle's restatements (bit-exact).
The OpenCV fixed-point paths themselves are PARITY UNPINNED (no cv2 anywhere), see DESIGN.md section 2."""
import numpy as np
import pytest

from metdetpy_amd.imgproc import Preprocessor, max_stack
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
