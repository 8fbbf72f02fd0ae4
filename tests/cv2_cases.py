"""Seeded inputs of the OpenCV pinning fixtures (tests/golden/g9_cv2_*.npz) -- NumPy only, so that tools/pin_cv2.py (which runs
wherever `cv2` imports) and the tests that consume its output (which run where it does not) build byte-identical inputs.

Every OpenCV call site of the hot path is covered:
    cv2.medianBlur(.,3)   Detector.py:329        cv2.threshold(BINARY)  Detector.py:332, 267-287
    cv2.morphologyEx      Detector.py:335        cv2.erode / dilate     Detector.py:241, 270-288
    cv2.absdiff           Detector.py:266-284    cv2.HoughLinesP        Detector.py:347-353, 289-293
    cv2.cvtColor          imgproc.py:87-88       cv2.resize             imgproc.py:82-85, fileio.py:250-292
"""
import functools

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


# threshold, minLineLength, maxLineGap.  2.5 / 3.5 / 0.5: HoughLinesP takes cvRound(maxLineGap), which rounds halves to EVEN
# (2, 4, 0) -- and M3Detector hands it a float (Detector.py:343-353: max(0, 1 - dst_sum / 0.05) * max_gap), so the tie rule is
# live on the hot path; the "dashes_halves" image below has dashes 2, 3 and 4 pixels apart, so each rounding gives other segments
HOUGH_PARAMS = [(10, 10, 10.0), (10, 15, 8.0), (10, 10, 0.0), (25, 30, 3.0), (1, 0, 0.0), (10, 10, 4.37), (10, 10, 2.5), (10, 10, 3.5),
                (10, 10, 0.5)]


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
    img = new(80, 360)
    for row, gap in ((10, 1), (25, 2), (40, 3), (55, 4), (70, 5)):
        x = 3
        while x + 7 < 360:
            img[row, x:x + 7] = 255
            x += 7 + gap
    out.append(("dashes_halves", img))
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


COLOR_SHAPES = [(2, 2), (67, 131), (270, 480), (216, 384), (540, 960), (4, 4096), (432, 768), (1080, 1920)]
RESIZES = {  # src (h, w) -> dsizes (w, h)
    (67, 131): [(131, 67), (65, 33), (40, 21), (262, 134), (100, 100)],
    (270, 480): [(240, 135), (120, 68), (96, 54), (480, 270), (333, 187)],
    (216, 384): [(96, 54), (192, 108), (384, 216)],
    (540, 960): [(480, 270), (240, 135), (958, 539)],
    (2, 2): [(1, 1), (4, 4), (3, 5)],
    (4, 4096): [(1024, 1)],      # the contested triples through a 4:1 reduction as well
    (432, 768): [(192, 108)],    # 4:1 in both directions, the reduction of config 1 (3840x2160 -> resize 960)
    (1080, 1920): [(960, 540)],  # 2:1 at full size: a 1080p source under the reference's default `resize: 960` (round 6)
}


@functools.lru_cache(maxsize=1)
def contested_gray_triples():
    """All (b, g, r) with b a multiple of 2 for which the two candidate coefficient sets of cv2.cvtColor(BGR2GRAY) on u8 disagree:
    15-bit (3735, 19235, 9798; + 2^14 >> 15 -- OpenCV 4.x's RGB2Gray<uchar>, what oracle/m3_oracle.py and the HIP kernel use) vs
    14-bit (1868, 9617, 4899; + 2^13 >> 14 -- OpenCV 2.x / 3.x, and what SURVEY App. A.12 recalled).  A pinned answer on these
    pixels says which one a given OpenCV build runs; an image of random pixels could agree with both by accident."""
    v = np.arange(0, 256, dtype=np.int32)
    b, g, r = np.meshgrid(v[::2], v, v, indexing="ij")
    g15 = (b * 3735 + g * 19235 + r * 9798 + 16384) >> 15
    g14 = (b * 1868 + g * 9617 + r * 4899 + 8192) >> 14
    sel = np.flatnonzero((g15 != g14).ravel())
    tri = np.stack([b.ravel()[sel], g.ravel()[sel], r.ravel()[sel]], axis=1).astype(np.uint8)
    return tri


def color_inputs():
    rng = np.random.default_rng(9003)
    out = []
    for (h, w) in COLOR_SHAPES:
        if (h, w) == (4, 4096):  # every pixel is one on which the 15-bit and the 14-bit gray coefficients disagree
            tri = contested_gray_triples()
            assert len(tri) >= h * w, len(tri)
            out.append((f"bgr_{h}x{w}", np.ascontiguousarray(tri[:h * w].reshape(h, w, 3))))
            continue
        if (h, w) == (1080, 1920):
            # full size, but smooth (sinusoids + a ramp + 0.5 % salt): the four pinned arrays of this shape compress to a few hundred KB
            yy, xx = np.mgrid[:h, :w].astype(np.float32)
            img = np.stack([(np.sin(xx / 23.0) * np.cos(yy / 31.0) + 1) * 110 + 10, (xx + 2 * yy) % 251,
                            (np.cos((xx - yy) / 17.0) + 1) * 90 + 40], axis=-1).astype(np.uint8)
            salt = rng.random((h, w)) < 0.005
            img[salt] = rng.integers(0, 256, (int(salt.sum()), 3), dtype=np.uint8)
            out.append((f"bgr_{h}x{w}", img))
            continue
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if h >= 60:  # some smooth structure as well, so that the bilinear taps matter
            yy, xx = np.mgrid[:h, :w]
            img[..., 1] = ((np.sin(xx / 7.0) + np.cos(yy / 5.0) + 2) * 60).astype(np.uint8)
        out.append((f"bgr_{h}x{w}", img))
    return out


CLASSIC = dict(H=135, W=240, frames=12, seed=9004, thresholds=[12, 20], hough=(10, 10, 10))


def classic_frames():
    """A small sky with a fast streak for the ClassicDetector chain (Detector.py:257-296: four frames, absdiff / threshold / dilate /
    255 - x / bitwise_and / HoughLinesP)."""
    from metdetpy_amd.synth import SkyStream
    C = CLASSIC
    return SkyStream(C["H"], C["W"], C["seed"], speed=9, streak_period=6, streak_life=5, n_stars=30).take(C["frames"])


M3_STREAM = dict(H=270, W=480, n=8, seed=20220413, frames=70, speed=10, streak_period=30, sensitivities=["normal", "high", "low"])
# the reference's default window (config/m3det_normal.json: window_sec 1 at 25 frames per second -> n = 25: the window lengths above 16
# have window kernels of their own since round 6); keys carry the prefix "n25_"
M3_STREAM_LONG = dict(H=192, W=320, n=25, seed=20220430, frames=95, speed=6, streak_period=40, sensitivities=["normal"], prefix="n25_")
