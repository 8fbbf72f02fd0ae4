"""Shared helpers for the parity tests."""
import numpy as np

from metdetpy_amd.metstruct import BinaryCfg, BinaryCoreCfg, DynamicCfg, HoughLineCfg


def binary_cfg(sensitivity="normal", adaptive=True, init_value=7, area=0.1, interval=2, hough=(10, 10, 10), dy_mask=True):
    return BinaryCfg(BinaryCoreCfg(adaptive, init_value, sensitivity, area, interval), HoughLineCfg(*hough),
                     DynamicCfg(dy_mask, 5))


def oracle_kwargs(cfg: BinaryCfg):
    return dict(adaptive=cfg.binary.adaptive_bi_thre, init_value=cfg.binary.init_value, sensitivity=cfg.binary.sensitivity,
                area=cfg.binary.area, interval=cfg.binary.interval, hough_threshold=cfg.hough_line.threshold,
                hough_min_len=cfg.hough_line.min_len, hough_max_gap=cfg.hough_line.max_gap, dy_mask=cfg.dynamic.dy_mask)


def frames_normal(seed, count, shape, mu, sigma):
    rng = np.random.default_rng(seed)
    return [np.clip(np.rint(rng.normal(mu, sigma, shape)), 0, 255).astype(np.uint8) for _ in range(count)]


def busy_frames(seed, count, shape, density=0.02):
    """Frames that light up lots of structure: noise + random bright segments that move, appear and persist,
    so the median/close/dy-mask/Hough stages all have work on every frame."""
    rng = np.random.default_rng(seed)
    H, W = shape
    out = []
    static = np.zeros(shape, np.float32)
    for _ in range(6):
        y, x = rng.integers(2, H - 2), rng.integers(2, W - 12)
        static[y:y + 2, x:x + 10] += 60
    for t in range(count):
        img = rng.normal(30, 2.5, shape).astype(np.float32)
        if t > count // 3:
            img += static  # appears late and then persists -> exercised by the dynamic mask
        for _ in range(int(rng.integers(1, 5))):
            y0, x0 = rng.integers(0, H), rng.integers(0, W)
            ang = rng.uniform(0, np.pi)
            ln = int(rng.integers(8, max(9, min(H, W) // 2)))
            for s in range(ln):
                yy, xx = int(y0 + s * np.sin(ang)), int(x0 + s * np.cos(ang))
                if 0 <= yy < H and 0 <= xx < W:
                    img[max(0, yy - 1):yy + 1, max(0, xx - 1):xx + 1] += 70
        img += (rng.random(shape) < density) * rng.uniform(20, 80)
        out.append(np.clip(np.rint(img), 0, 255).astype(np.uint8))
    return out
