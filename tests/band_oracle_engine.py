"""A band engine made of the CPU oracle's functions -- TEST INFRASTRUCTURE.

`metdetpy_amd.bands` drives an engine through three calls per pass (noise -> pixels -> hough_stage).  The product engine
is the HIP library; this one does the same arithmetic with `oracle/` so that the band tiling itself -- halo size, ROI
split, all-reduce of the noise sums, the all-to-all layout, frame-to-rank assignment, the result gather -- can be
checked against the whole-frame oracle on CPU, including with real multi-process gloo collectives."""
import numpy as np

from metdetpy_amd import _lib
from oracle import c_oracle as co
from oracle import m3_oracle as mo


def pack_rows(img255: np.ndarray, WW: int) -> np.ndarray:
    """(rows, W) u8 {0,255} -> (rows, WW) u64, bit k of word w = pixel 64w+k."""
    rows, W = img255.shape
    bits = np.zeros((rows, WW * 64), np.uint8)
    bits[:, :W] = img255 != 0
    return np.packbits(bits, axis=1, bitorder="little").view(np.uint64)


def unpack_rows(words: np.ndarray, W: int) -> np.ndarray:
    bits = np.unpackbits(np.ascontiguousarray(words).view(np.uint8), axis=1, bitorder="little")
    return (bits[:, :W] * 255).astype(np.uint8)


class OracleBandEngine:

    def __init__(self, plan, b, n, cfg, mask_area, max_batch, device):
        self.plan, self.b, self.n, self.cfg = plan, b, n, cfg
        d = plan.bands[b]
        self.d = d
        self.shape = (d["sub1"] - d["sub0"], plan.W)
        self.mask_area = mask_area
        self.ring = mo.RingWindow(n, self.shape)
        self.dy_ring = mo.RingWindow(n, self.shape)
        r = d["roi"]
        self.roi = r
        self.roi_ring = mo.RingWindow(n, (r[2] - r[0], r[3] - r[1]))
        self.ema = mo.WarmupEMA(1 - cfg.binary.interval / 60, n)
        self.std_interval = cfg.binary.interval * n
        self.t = 0
        self.thr = mo.SENSITIVITY_INIT[cfg.binary.sensitivity] if cfg.binary.adaptive_bi_thre else cfg.binary.init_value
        self.thr_f, self.sigma = float(self.thr), 0.0
        self._staged = None

    def _due(self, t):
        if t > self.n and t % self.std_interval == 0:
            return self.n
        if 1 < t <= self.n:
            return t
        return 0

    def noise(self, sub_frames):
        assert sub_frames.shape[1:] == self.shape
        r, out, t = self.roi, [], self.t
        for f in sub_frames:
            t += 1
            self.roi_ring.push(f[r[0]:r[2], r[1]:r[3]])
            k = self._due(t)
            if k:
                s1, s2, _ = mo.noise_sums(self.roi_ring.ring, self.roi_ring.mean, k) if self.roi_ring.ring.size else (0, 0, 0)
                out += [s1, s2]
        self._staged = sub_frames
        return np.array(out, np.int64)

    def pixels(self, sums_total, out=None):
        assert out is None
        p, d, N = self.plan, self.d, self.plan.nbands
        K = len(self._staged)
        kp = -(-K // N)
        send = np.zeros((N, kp, p.band_rows, p.WW), np.uint64)
        info = np.zeros(K, _lib.FRAME_INFO_DTYPE)
        a, bcoef = mo.SENSITIVITY_COEF[self.cfg.binary.sensitivity]
        job = 0
        for i, f in enumerate(self._staged):
            self.t += 1
            k = self._due(self.t)
            if k:
                self.sigma = mo.sigma_from_sums(int(sums_total[2 * job]), int(sums_total[2 * job + 1]), k * p.roi_px)
                self.ema.update(self.sigma)
                job += 1
                if self.cfg.binary.adaptive_bi_thre and self.ema.value != 0:
                    self.thr_f = a * self.ema.value**2 + bcoef
                    self.thr = round(self.thr_f)
            self.ring.push(f)
            diff = mo.median_blur3(self.ring.peak - self.ring.mean)
            act = mo.morph_close3(mo.threshold_binary(diff, self.thr, 255))
            dst = act
            if self.cfg.dynamic.dy_mask:
                self.dy_ring.push(act)
                alive = np.array(self.dy_ring.total <= (self.dy_ring.length - 1) * 255, dtype=np.uint8)
                dst = np.multiply(act, mo.erode3(alive))
            own = dst[d["own0"] - d["sub0"]:d["own1"] - d["sub0"]]
            send[i % N, i // N, :len(own)] = pack_rows(own, p.WW)
            info[i]["bi_threshold"], info[i]["bi_threshold_float"] = self.thr, self.thr_f
            info[i]["snr"], info[i]["sigma"] = self.ema.value, self.sigma
        self._staged = None
        return send, info

    def hough_stage(self, pieces, kp, on_device=False):
        assert not on_device
        p = self.plan
        pieces = np.asarray(pieces).reshape(p.nbands, kp, p.band_rows, p.WW)
        info = np.zeros(kp, _lib.FRAME_INFO_DTYPE)
        raw = np.zeros((kp, _lib.MDH_MAX_LINES, 4), np.int32)
        nms = np.zeros((kp, _lib.MDH_MAX_LINES, 4), np.int32)
        prob = np.zeros((kp, _lib.MDH_MAX_LINES), np.float64)
        thr, min_len, max_gap = self.cfg.hough_line.threshold, self.cfg.hough_line.min_len, self.cfg.hough_line.max_gap
        for j in range(kp):
            plane = np.concatenate([pieces[b, j] for b in range(p.nbands)], axis=0)[:p.H]
            dst = unpack_rows(plane, p.W)
            lit = int(np.count_nonzero(dst))
            dst_sum = float(np.sum(dst / 255.) / self.mask_area * 100)  # Detector.py:342
            gap = max(0, 1 - dst_sum / 0.05) * max_gap                   # Detector.py:343-344
            lines = co.hough_lines_p(dst, 1, np.pi / 180.0, thr, min_len, gap)
            info[j]["n_raw"], info[j]["lit_count"], info[j]["dst_sum"], info[j]["hough_gap"] = len(lines), lit, dst_sum, gap
            raw[j, :min(len(lines), _lib.MDH_MAX_LINES)] = lines[:_lib.MDH_MAX_LINES]
            if 0 < len(lines) <= 500:
                kept, nonline = mo.lineset_nms(lines)
                info[j]["n_nms"] = len(kept)
                nms[j, :len(kept)], prob[j, :len(kept)] = kept, nonline
        return info, raw, nms, prob

    def skip(self):
        """Drop the staged pass: the frames stay in the noise window, nothing else happens."""
        self.t += len(self._staged)
        self._staged = None

    def seek(self, t0, ema_updates=0, snr=0.0, sigma=0.0, thr_float=0.0, thr=0):
        r = self.roi
        self.ring = mo.RingWindow(self.n, self.shape)
        self.dy_ring = mo.RingWindow(self.n, self.shape)
        self.roi_ring = mo.RingWindow(self.n, (r[2] - r[0], r[3] - r[1]))
        self.t = int(t0)
        self.ema = mo.WarmupEMA(1 - self.cfg.binary.interval / 60, self.n)
        self.ema.t, self.ema.value = int(ema_updates), snr
        self.thr, self.thr_f, self.sigma = int(thr), thr_float, sigma

    def close(self):
        pass
