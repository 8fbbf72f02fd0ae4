"""Seeded inputs of the OpenCV pinning fixtures (tests/golden/g9_cv2_*.npz) -- NumPy only, so that tools/pin_cv2.py (which runs
wherever `cv2` imports) and the tests that consume its output (which run where it does not) build byte-identical inputs.

Every OpenCV call site of the hot path is covered:
    cv2.medianBlur(.,3)   Detector.py:329        cv2.threshold(BINARY)  Detector.py:332, 267-287
    cv2.morphologyEx      Detector.py:335        cv2.erode / dilate     Detector.py:241, 270-288
    cv2.absdiff           Detector.py:266-284    cv2.HoughLinesP        Detector.py:347-353, 289-293
    cv2.cvtColor          imgproc.py:87-88       cv2.resize             imgproc.py:82-85, fileio.py:250-292
"""
import numpy as np

STENCIL_SHAPES = [(1, 1), (1, 7), (7, 1), (2, 2), (5, 9), (37, 53), (64, 64), (135, 241)]
THRESHOLDS = [0, 5, 128, 254, 255]
DENSITIES = [0.02, 0.3, 0.7]


def stencil_inputs():
    """[(name, u8 image)]: noise-like gray images for median / threshold / absdiff, binary images for the morphology."""
    rng = np.random.default_rng(9001)
    out = []
    for (h, w) in STENCIL_SHAPES:
        out.append((f"gray_{h}x{w}", np.clip(np.rint(rng.normal(30, 12, (h, w))), 0, 255).astype(np.uint8)))
        out.append((f"grayb_{h}x{w}", rng.integers(0, 256, (h, w), dtype=np.uint8)))
        for d in DENSITIES:
            out.append((f"bin{int(d * 100)}_{h}x{w}", ((rng.random((h, w)) < d) * 255).astype(np.uint8)))
    return out


def _draw_line(img, x0, y0, x1, y1, width=1):
    n = int(max(abs(x1 - x0), abs(y1 - y0))) + 1
    xs = np.rint(np.linspace(x0, x1, n)).astype(int)
    ys = np.rint(np.linspace(y0, y1, n)).astype(int)
    for d in range(width):
        ok = (xs >= 0) & (xs < img.shape[1]) & (ys + d >= 0) & (ys + d < img.shape[0])
        img[ys[ok] + d, xs[ok]] = 255


HOUGH_PARAMS = [(10, 10, 10.0), (10, 15, 8.0), (10, 10, 0.0), (25, 30, 3.0), (1, 0, 0.0), (10, 10, 4.37)]  # threshold, minLineLength, maxLineGap


def hough_inputs():
    """[(name, binary image)]: line fans, borders, dashes around the gap limit, noise, blobs, streak-like thick segments."""
    rng = np.random.default_rng(9002)
    out = []

    def new(h, w):
        return np.zeros((h, w), np.uint8)

    img = new(200, 320)
    for a in np.linspace(0, np.pi, 13, endpoint=False):
        r = 90
        _draw_line(img, 160 - r * np.cos(a), 100 - r * np.sin(a), 160 + r * np.cos(a), 100 + r * np.sin(a))
    out.append(("fan", img))
    img = new(97, 131)
    img[0, :] = img[-1, :] = 255
    img[:, 0] = img[:, -1] = 255
    _draw_line(img, 0, 0, 130, 96)
    out.append(("borders", img))
    img = new(90, 400)
    for row, gap in ((20, 3), (45, 9), (70, 12)):
        x = 2
        while x + 12 < 400:
            img[row, x:x + 12] = 255
            x += 12 + gap
    out.append(("dashed", img))
    img = new(270, 480)
    img[rng.random(img.shape) < 0.01] = 255
    out.append(("noise", img))
    img = new(120, 200)
    img[rng.random(img.shape) < 0.12] = 255
    out.append(("dense", img))
    img = new(300, 500)
    for _ in range(40):
        x, y = rng.uniform(0, 500), rng.uniform(0, 300)
        a, ln = rng.uniform(0, np.pi), rng.uniform(5, 60)
        _draw_line(img, x, y, x + ln * np.cos(a), y + ln * np.sin(a), width=int(rng.integers(1, 3)))
    img[rng.random(img.shape) < 0.002] = 255
    out.append(("segments", img))
    img = new(128, 128)
    yy, xx = np.ogrid[:128, :128]
    for _ in range(12):
        x, y, r = int(rng.uniform(0, 128)), int(rng.uniform(0, 128)), int(rng.integers(2, 9))
        img[(yy - y)**2 + (xx - x)**2 <= r * r] = 255
    out.append(("blobs", img))
    img = new(540, 960)
    for k in range(3):  # streak-like: 3-4 px wide, ~200 px long, arbitrary angles
        a = rng.uniform(0, np.pi)
        x, y = rng.uniform(250, 700), rng.uniform(150, 400)
        _draw_line(img, x, y, x + 200 * np.cos(a), y + 200 * np.sin(a), width=3 + k % 2)
    out.append(("streaks", img))
    out.append(("single", np.pad(np.full((1, 1), 255, np.uint8), ((16, 16), (23, 46)))))
    out.append(("empty", new(64, 64)))
    img = new(1, 300)
    img[0, 20:180] = 255
    out.append(("one_row", img))
    img = new(300, 1)
    img[40:260, 0] = 255
    out.append(("one_col", img))
    return out


COLOR_SHAPES = [(2, 2), (67, 131), (270, 480), (216, 384), (540, 960)]
RESIZES = {  # src (h, w) -> dsizes (w, h)
    (67, 131): [(131, 67), (65, 33), (40, 21), (262, 134), (100, 100)],
    (270, 480): [(240, 135), (120, 68), (96, 54), (480, 270), (333, 187)],
    (216, 384): [(96, 54), (192, 108), (384, 216)],
    (540, 960): [(480, 270), (240, 135), (958, 539)],
    (2, 2): [(1, 1), (4, 4), (3, 5)],
}


def color_inputs():
    rng = np.random.default_rng(9003)
    out = []
    for (h, w) in COLOR_SHAPES:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if h >= 60:  # some smooth structure as well, so that the bilinear taps matter
            yy, xx = np.mgrid[:h, :w]
            img[..., 1] = ((np.sin(xx / 7.0) + np.cos(yy / 5.0) + 2) * 60).astype(np.uint8)
        out.append((f"bgr_{h}x{w}", img))
    return out


M3_STREAM = dict(H=270, W=480, n=8, seed=20220413, frames=70, speed=10, streak_period=30, sensitivities=["normal", "high", "low"])
