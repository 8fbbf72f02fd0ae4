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
                 mask01: np.ndarray | None = None):
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
