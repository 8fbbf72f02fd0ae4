"""Seeded synthetic night-sky streams -- NumPy only, identical on every machine.

Background noise clip(round(N(mu, sigma))), fixed 3x3 "stars", and one meteor-like streak every
`streak_period` frames: a 3-px-wide anti-aliased segment, +60 over the background, advancing
`speed` px per frame along a random direction for `streak_life` frames.  Uniform-random frames are
useless for this detector (every pixel fires and the frame is discarded), hence this generator.
"""
from __future__ import annotations

import numpy as np

SEEDS = {"C2": 20220413, "C3": 20220414, "C5": 20220415}


class SkyStream:

    def __init__(self, H: int, W: int, seed: int, *, mu: float = 30.0, sigma: float = 2.5, n_stars: int = 200,
                 streak_period: int = 50, streak_life: int = 12, speed: float | None = None, streak_gain: float = 60.0,
                 mask01: np.ndarray | None = None, hot_frac: float = 0.0, cloud_period: int = 0, cloud_gain: float = 25.0,
                 cloud_width: int | None = None):
        """The unfriendly variants (bench.py --stream): `sigma` = 6 for a noisy sky; `hot_frac` of the pixels are hot (a fixed
        offset of +40 .. +160 and three times the noise: they flicker over the threshold on their own); every `cloud_period`-th
        frame carries a "cloud edge": a wavy strip `cloud_width` pixels wide across the whole frame, +`cloud_gain` for that one
        frame -- more than 8192 lit pixels in the binaries of the n frames whose windows hold it.  All drawn after the stars and
        before the streak, from the same generator, so the default stream (none of them) is unchanged."""
        self.H, self.W = H, W
        self.rng = np.random.default_rng(seed)
        self.mu, self.sigma = mu, sigma
        self.period, self.life = streak_period, streak_life
        self.speed = 24.0 * W / 1920.0 if speed is None else speed
        self.gain = streak_gain
        self.mask01 = mask01
        self.t = 0
        stars = np.zeros((H, W), np.float32)
        ys = self.rng.integers(1, H - 1, n_stars)
        xs = self.rng.integers(1, W - 1, n_stars)
        amp = self.rng.uniform(40, 150, n_stars)
        for y, x, a in zip(ys, xs, amp):
            stars[y - 1:y + 2, x - 1:x + 2] += a * np.array([[.3, .6, .3], [.6, 1, .6], [.3, .6, .3]], np.float32)
        self.stars = stars
        self._streak = None
        self.hot = None
        if hot_frac > 0:
            m = self.rng.random((H, W)) < hot_frac
            self.hot = (m, self.rng.uniform(40, 160, int(m.sum())).astype(np.float32))
        self.cloud_period, self.cloud_gain = int(cloud_period), float(cloud_gain)
        self.cloud_width = int(cloud_width) if cloud_width else max(3, -(-11520 // W))  # (> 8192 lit pixels whatever the width)

    def _new_streak(self):
        ang = self.rng.uniform(0, 2 * np.pi)
        v = np.array([np.cos(ang), np.sin(ang)]) * self.speed
        span = self.life * self.speed
        m = span + 8
        x0 = self.rng.uniform(m, max(m + 1, self.W - m))
        y0 = self.rng.uniform(m, max(m + 1, self.H - m))
        self._streak = (np.array([x0, y0]), v, 0)

    def _draw_streak(self, img):
        p0, v, k = self._streak
        a = p0 + v * k
        b = p0 + v * (k + 1)
        lo = np.floor(np.minimum(a, b) - 3).astype(int)
        hi = np.ceil(np.maximum(a, b) + 3).astype(int)
        x_lo, y_lo = max(lo[0], 0), max(lo[1], 0)
        x_hi, y_hi = min(hi[0], self.W - 1), min(hi[1], self.H - 1)
        if x_lo > x_hi or y_lo > y_hi:
            return
        yy, xx = np.mgrid[y_lo:y_hi + 1, x_lo:x_hi + 1]
        d = b - a
        tt = np.clip(((xx - a[0]) * d[0] + (yy - a[1]) * d[1]) / (d @ d), 0, 1)
        dist = np.hypot(xx - (a[0] + tt * d[0]), yy - (a[1] + tt * d[1]))
        img[y_lo:y_hi + 1, x_lo:x_hi + 1] += (self.gain * np.clip(2.0 - dist, 0, 1)).astype(np.float32)

    def next(self) -> np.ndarray:
        img = self.rng.standard_normal((self.H, self.W), dtype=np.float32) * np.float32(self.sigma) + np.float32(self.mu)
        img += self.stars
        if self.hot is not None:
            m, off = self.hot
            img[m] += off + self.rng.standard_normal(len(off), dtype=np.float32) * np.float32(2 * self.sigma)
        if self.cloud_period and self.t % self.cloud_period == self.cloud_period // 2:
            xs = np.arange(self.W)
            y0 = self.rng.uniform(0.25, 0.75) * self.H
            ys = np.rint(y0 + 0.08 * self.H * np.sin(xs / self.W * 2 * np.pi * self.rng.uniform(1, 3) + self.rng.uniform(0, 6))).astype(int)
            for d in range(self.cloud_width):
                yy = np.clip(ys + d, 0, self.H - 1)
                img[yy, xs] += np.float32(self.cloud_gain)
        if self.t % self.period == self.period // 5:
            self._new_streak()
        if self._streak is not None:
            self._draw_streak(img)
            p0, v, k = self._streak
            self._streak = (p0, v, k + 1) if k + 1 < self.life else None
        self.t += 1
        out = np.clip(np.rint(img), 0, 255).astype(np.uint8)
        if self.mask01 is not None:
            out *= self.mask01
        return out

    def take(self, k: int) -> np.ndarray:
        return np.stack([self.next() for _ in range(k)], axis=0)


class Config1Clip:
    """Stand-in for BASELINE config 1's video (test/20220413Red.mp4: a Git-LFS blob the reference tree does not hold), built from
    what the reference DOES hold about it: test/20220413_annotation.json (one meteor, 00:00:02.4 -> 00:00:04.4, pt1 [433, 147] ->
    pt2 [374, 222] at anno_size [1168, 655]) and test/clip_test.json (the same meteor at [1412, 496] -> [1232, 728], i.e. a
    3840x2160 source; its times are multiples of 0.04 s: 25 frames per second).  tests/golden/config1_truth.json carries those
    numbers as data.

    A 3840x2160 BGR clip at 25 fps: a night sky at the runtime resolution (noise, stars) blown up 4x with a fixed full-resolution
    dither on top and slightly different channel gains (so that BGR2GRAY and the 4:1 INTER_LINEAR resize have real work), and ONE
    meteor drawn at full resolution: its head travels pt1 -> pt2 (scaled from anno_size to the frame) at constant speed between
    start_time and end_time, dragging a trail of `trail_s` seconds that runs out after the head has stopped.  NumPy only, seeded.
    """

    def __init__(self, truth: dict, *, size_hw=(2160, 3840), fps: float = 25.0, duration_s: float = 6.0, seed: int = 20220413,
                 gain: float = 70.0, trail_s: float = 0.32, width_px: float = 6.0):
        self.H, self.W = size_hw
        self.fps, self.total = float(fps), int(round(duration_s * fps))
        ax, ay = truth["anno_size"]
        m = truth["meteors"][0]
        sx, sy = self.W / ax, self.H / ay
        self.p1 = np.array([m["pt1"][0] * sx, m["pt1"][1] * sy])
        self.p2 = np.array([m["pt2"][0] * sx, m["pt2"][1] * sy])
        self.t0, self.t1 = _clock_s(m["start_time"]), _clock_s(m["end_time"])
        self.gain, self.trail_s, self.width = float(gain), float(trail_s), float(width_px)
        self.small = SkyStream(self.H // 4, self.W // 4, seed, streak_period=10 ** 9)  # (no streaks of its own)
        rng = np.random.default_rng(seed + 7)
        self.dither = rng.integers(-3, 4, (self.H, self.W, 3), dtype=np.int16)
        self.chan_gain = np.array([0.92, 1.0, 1.06], np.float32)
        self.k = 0

    def _head(self, t: float) -> np.ndarray:
        u = min(max((t - self.t0) / (self.t1 - self.t0), 0.0), 1.0)
        return self.p1 + (self.p2 - self.p1) * u

    def next(self) -> np.ndarray:
        """The next source frame, (H, W, 3) uint8 BGR: clip(rint(sky * channel gain) + dither), sky = the small frame blown up 4x
        (+ the meteor where it is).  The gains are applied at the small resolution (the same values, a sixteenth of the work)."""
        t = self.k / self.fps
        base = self.small.next().astype(np.float32)
        small3 = np.rint(base[..., None] * self.chan_gain).astype(np.int16)
        img = np.repeat(np.repeat(small3, 4, 0), 4, 1)
        img += self.dither
        if self.t0 <= t < self.t1 + self.trail_s:
            a, b = self._head(t - self.trail_s), self._head(t)
            if t - self.trail_s < self.t0:
                a = self.p1
            pad = self.width + 2
            lo = np.floor(np.minimum(a, b) - pad).astype(int)
            hi = np.ceil(np.maximum(a, b) + pad).astype(int)
            x_lo, y_lo, x_hi, y_hi = max(lo[0], 0), max(lo[1], 0), min(hi[0], self.W - 1), min(hi[1], self.H - 1)
            yy, xx = np.mgrid[y_lo:y_hi + 1, x_lo:x_hi + 1]
            d = b - a
            dd = float(d @ d)
            tt = np.clip(((xx - a[0]) * d[0] + (yy - a[1]) * d[1]) / dd, 0, 1) if dd > 0 else np.zeros(xx.shape)
            dist = np.hypot(xx - (a[0] + tt * d[0]), yy - (a[1] + tt * d[1]))
            # brightest at the head, fading along the trail
            add = (self.gain * (0.45 + 0.55 * tt) * np.clip(self.width / 2 + 1 - dist, 0, 1)).astype(np.float32)
            sky = base[yy // 4, xx // 4] + add
            img[y_lo:y_hi + 1, x_lo:x_hi + 1] = (np.rint(sky[..., None] * self.chan_gain).astype(np.int16)
                                                 + self.dither[y_lo:y_hi + 1, x_lo:x_hi + 1])
        self.k += 1
        return np.clip(img, 0, 255).astype(np.uint8)


def _clock_s(text: str) -> float:
    h, m, s = text.split(":")
    return int(h) * 3600 + int(m) * 60 + float(s)
