"""One stream tiled into row bands across GPUs (BASELINE config 5: a 7680x4320 stream over 8 MI355X; SURVEY 8e
"spatial row-bands").

Why it shards this way.  Everything `M3Detector.detect` does before `cv2.HoughLinesP` (Detector.py:327-340) is a
chain of pointwise ops and 3x3 stencils -- median, dilate, erode, and the dynamic mask's erode (Detector.py:234-242) --
so a band of rows plus MDH_BAND_HALO = 4 rows of each neighbour can be processed as if it were a frame of its own: the
artificial borders spoil at most the halo rows, which are thrown away.  What is global:

  * the noise sigma (Detector.py:81-91): sums over the sub-ROI, which spans several bands  -> all-reduce of
    (S1, S2) per due frame, two int64 each; every rank then runs the same EMA / threshold chain,
  * the Hough transform: sequential per frame (OpenCV's progressive probabilistic Hough), but frames are independent
    -> the dst bit planes (1 bit per pixel) go through ONE all-to-all per pass so that rank r holds the whole plane
    of frames r, r + world, ... and runs their Hough stage; the lit-pixel count and the adaptive gap
    (Detector.py:342-344) are computed there from the assembled plane, so they need no collective of their own,
  * the results: gathered to rank 0, which owns the MeteorCollector.

Collectives: RCCL over xGMI, enqueued through ctypes on the detector library's own streams (rccl.py) -- the band kernels
write straight into the send buffer in HBM and the Hough context reads the receive buffer in place; nothing of torch is
involved.  Ranks that have no communicator (several ranks sharing one device on a one-GPU box, or an engine substituted by
the CPU tests) exchange host buffers through the `group` object instead (dist.Group over the rendezvous store; the tests also
plug a gloo-backed group of their own in: tests/gloo_group.py).  `LocalBands` runs all bands of a frame in one process on one
GPU (same code path minus the collectives) for single-GPU parity tests.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib
from .metstruct import BinaryCfg
from .utils import select_std_roi

HALO = _lib.MDH_BAND_HALO


class BandPlan:
    """Geometry of the tiling: band b owns rows [own0, own1) and works on rows [sub0, sub1) (owned + halo)."""

    def __init__(self, H: int, W: int, nbands: int, std_roi):
        if nbands < 1:
            raise ValueError("nbands must be >= 1")
        self.H, self.W, self.nbands = int(H), int(W), int(nbands)
        self.WW = (self.W + 63) // 64
        self.band_rows = -(-self.H // self.nbands)
        if (self.nbands - 1) * self.band_rows >= self.H:
            raise ValueError(f"{H} rows cannot be cut into {nbands} non-empty bands of {self.band_rows} rows")
        self.std_roi = tuple(int(v) for v in std_roi)
        r0, c0, r1, c1 = self.std_roi
        self.roi_px = (r1 - r0) * (c1 - c0)
        self.bands = []
        for b in range(self.nbands):
            own0 = b * self.band_rows
            own1 = min(self.H, own0 + self.band_rows)
            sub0, sub1 = max(0, own0 - HALO), min(self.H, own1 + HALO)
            q0, q1 = max(r0, own0), min(r1, own1)  # rows of the noise ROI this band owns
            roi = (q0 - sub0, c0, q1 - sub0, c1) if q1 > q0 else (0, c0, 0, c1)
            self.bands.append(dict(own0=own0, own1=own1, sub0=sub0, sub1=sub1, roi=roi))

    def crop(self, frames: np.ndarray, b: int) -> np.ndarray:
        """Rows band b works on, of (K,H,W) frames (or of one (H,W) image)."""
        d = self.bands[b]
        return np.ascontiguousarray(frames[..., d["sub0"]:d["sub1"], :])


class HipBandEngine:
    """The product engine: one MDH_ROLE_BAND context on this rank's sub-image and one MDH_ROLE_HOUGH context on the whole
    frame, both on `device`.  (Tests may substitute an engine with the same three methods to exercise the exchange
    logic without a GPU.)"""

    def __init__(self, plan: BandPlan, b: int, n: int, cfg: BinaryCfg, mask_area: float, max_batch: int, device: int,
                 with_hough: bool = True):
        d = plan.bands[b]
        self.plan, self.b = plan, b

        def base(role, roi, kmax):
            c = _lib.MdhCfg()
            c.detector = _lib.MDH_DET_M3
            c.adaptive_bi_thre = int(bool(cfg.binary.adaptive_bi_thre))
            c.init_value = int(cfg.binary.init_value)
            c.sensitivity = _lib.SENSITIVITY_ID[cfg.binary.sensitivity]
            c.interval = int(cfg.binary.interval)
            for k in range(4):
                c.roi[k] = int(roi[k])
            c.hough_threshold = int(cfg.hough_line.threshold)
            c.hough_min_len = int(cfg.hough_line.min_len)
            c.hough_max_gap = int(cfg.hough_line.max_gap)
            c.dy_mask = int(bool(cfg.dynamic.dy_mask))
            c.max_batch = int(kmax)
            c.role = role
            c.mask_area = float(mask_area)
            return c

        self.band = _lib.Context(d["sub1"] - d["sub0"], plan.W, n, base(_lib.MDH_ROLE_BAND, d["roi"], max_batch), device=device)
        self.kp_max = -(-int(max_batch) // plan.nbands)
        self._hough_cfg = lambda kmax: base(_lib.MDH_ROLE_HOUGH, (0, 0, 0, 0), kmax)
        self.hough = _lib.Context(plan.H, plan.W, 1, self._hough_cfg(self.kp_max), device=device) if with_hough else None
        self.device = device

    def shared_hough_context(self, max_batch):
        """ONE Hough context for all frames of a pass (all bands in one process: LocalBands)."""
        return _lib.Context(self.plan.H, self.plan.W, 1, self._hough_cfg(int(max_batch)), device=self.device)

    def noise(self, sub_frames, K=None, on_device=False):
        return self.band.band_noise(sub_frames, K=K, on_device=on_device)

    def pixels(self, sums_total, out=None, nsplit=None):
        d = self.plan.bands[self.b]
        return self.band.band_pixels(sums_total, self.plan.roi_px, d["own0"] - d["sub0"], d["own1"] - d["own0"],
                                     self.plan.band_rows, nsplit or self.plan.nbands, out=out)

    def hough_stage(self, pieces, kp, on_device=False):
        return self.hough.hough_planes(pieces, kp, self.plan.nbands, self.plan.band_rows, on_device=on_device)

    def hough_submit(self, pieces, kp, on_device=False):
        """hough_stage without the wait (mdh_hough_planes_submit): the replay of this pass runs under the band stages of the next."""
        self.hough.hough_planes_submit(pieces, kp, self.plan.nbands, self.plan.band_rows, on_device=on_device)

    def hough_collect(self, kp):
        return self.hough.hough_planes_collect(kp)

    # ---- queued forms: nothing below waits for the GPU (include/metdet_hip.h, "WITHOUT host round trips") ----
    def noise_submit(self, sub_frames, K=None, on_device=False) -> int:
        return self.band.band_submit(sub_frames, K=K, on_device=on_device)

    def pixels_submit(self, out: int, nsplit=None):
        d = self.plan.bands[self.b]
        self.band.band_pixels_submit(self.plan.roi_px, d["own0"] - d["sub0"], d["own1"] - d["own0"], self.plan.band_rows,
                                     nsplit or self.plan.nbands, out)

    def pixels_submit_into_planes(self, planes: int):
        """Owned rows straight into whole-frame planes [K][H][WW] (the shared Hough context's input: mdh_hough_input_planes)."""
        d = self.plan.bands[self.b]
        self.band.band_pixels_submit(self.plan.roi_px, d["own0"] - d["sub0"], d["own1"] - d["own0"], self.plan.H, 1,
                                     planes + d["own0"] * self.plan.WW * 8, nofill=True)

    def thresholds_collect(self):
        return self.band.band_collect()

    def skip(self):
        self.band.band_skip()

    def seek(self, t0, **state):
        self.band.seek(t0, **state)

    def close(self):
        self.band.close()
        if self.hough is not None:
            self.hough.close()


def _package(num_cls, info, raw, nms, prob):
    """(lines, cls_pred) as M3Detector.detect returns them (Detector.py:358-392)."""
    n_raw, n_nms = int(info["n_raw"]), int(info["n_nms"])
    if n_raw == 0 or n_raw > 500:
        return np.array([]), np.zeros((0, num_cls))
    pred = np.zeros((n_nms, num_cls))
    pred[:, -1] = prob[:n_nms]
    pred[:, 0] = 1 - prob[:n_nms]
    return nms[:n_nms].copy(), pred


class _BandSetup:
    """What every driver derives from the detector arguments (same constructor as LineDetector, Detector.py:186-220)."""

    def __init__(self, window_sec, fps, mask, num_cls, cfg: BinaryCfg, nbands):
        self.num_cls = num_cls
        self.n = int(window_sec * fps)
        if not 1 <= self.n <= _lib.MDH_MAX_WINDOW:
            raise ValueError(f"window of {self.n} frames unsupported (1..{_lib.MDH_MAX_WINDOW})")
        mask = np.asarray(mask)
        H, W = mask.shape[:2]
        self.mask_area = float(np.sum(mask))
        self.plan = BandPlan(H, W, nbands, select_std_roi(mask, cfg.binary.area))
        self.cfg = cfg


def _merge_info(thr_info, hough_info):
    out = hough_info.copy()
    for f in ("bi_threshold", "bi_threshold_float", "snr", "sigma"):
        out[f] = thr_info[f]
    return out


class LocalBands:
    """All bands of one stream in ONE process on ONE GPU: the same three steps per band, the exchanges done with array
    copies.  For single-GPU parity tests and for sizing the band path; production runs one BandedM3Detector per GPU."""

    def __init__(self, window_sec, fps, mask, num_cls, cfg, nbands, *, device=0, max_batch=64, engine=HipBandEngine,
                 shared_hough=None):
        """shared_hough (default: yes with the HIP engine): the bands of one process hand their rows to ONE Hough context that
        replays every frame of the pass -- each band's export kernel writes its piece of every plane straight into that context's
        input buffer, so the exchange costs nothing -- instead of one Hough context per band with an all-to-all in between (the
        multi-GPU structure, kept for the tests of the exchange logic: shared_hough=False)."""
        self.s = _BandSetup(window_sec, fps, mask, num_cls, cfg, nbands)
        p = self.s.plan
        if engine is HipBandEngine and nbands > 2:
            _lib.prefer_hw_queues(4)  # many contexts on one device: few hardware queues (no effect once the runtime is up)
        self.shared = (engine is HipBandEngine) if shared_hough is None else bool(shared_hough)
        if self.shared:
            self.engines = [engine(p, b, self.s.n, cfg, self.s.mask_area, max_batch, device, with_hough=False) for b in range(p.nbands)]
            self.hough = self.engines[0].shared_hough_context(max_batch)
            # MDH_BANDS_SHARE_STREAMS (default 1): the bands of one process queue their stages on band 0's streams
            # (mdh_share_streams).  On by default since round 4: with every band's scalar chain queued before the first band's
            # window kernel (mdh_band_thresholds_submit) 8 bands of an 8K stream run at 61-73 k frames/s shared against 51-59 k with
            # a set of streams per band (sharing ALONE, without that reordering, had been slower: 30 k against 52 k;
            # profiles/EXPERIMENTS.md, round 4).  0 = a set of streams per band, for A/B.
            if os.environ.get("MDH_BANDS_SHARE_STREAMS", "1") != "0":
                for e in self.engines[1:]:
                    e.band.share_streams_of(self.engines[0].band)
        else:
            self.engines = [engine(p, b, self.s.n, cfg, self.s.mask_area, max_batch, device) for b in range(p.nbands)]
            self.hough = None
        # one host thread per band: every band has a context of its own (a context is used by one thread at a time, and ctypes
        # releases the GIL inside a call), so the bands' blocking steps run side by side on the device the way they do on N GPUs
        from concurrent.futures import ThreadPoolExecutor
        self._pool = ThreadPoolExecutor(max_workers=p.nbands) if engine is HipBandEngine and p.nbands > 1 else None

    def _each(self, fn):
        """fn(b, engine) for every band; side by side when the engines are the HIP ones."""
        if self._pool is None:
            return [fn(b, e) for b, e in enumerate(self.engines)]
        return list(self._pool.map(lambda be: fn(*be), enumerate(self.engines)))

    def detect_batch(self, frames: np.ndarray):
        """frames: (K,H,W) u8, whole frames.  Returns (results, info) like M3Detector.detect_batch."""
        p = self.s.plan
        K, N = len(frames), p.nbands
        kp = -(-K // N)
        parts = [e.noise(p.crop(frames, b)) for b, e in enumerate(self.engines)]
        total = np.sum(parts, axis=0) if len(parts[0]) else parts[0]           # "all-reduce"
        info = np.zeros(K, _lib.FRAME_INFO_DTYPE)
        results = [None] * K
        if self.shared:
            sent = [e.pixels(total, nsplit=1) for e in self.engines]           # [(1, K, band_rows, WW), info] per band: frame order
            hi, raw, nms, prob = self.hough.hough_planes(np.stack([x[0][0] for x in sent]), K, N, p.band_rows)
            for i in range(K):
                info[i] = _merge_info(sent[0][1][i], hi[i])
                results[i] = _package(self.s.num_cls, hi[i], raw[i], nms[i], prob[i])
            return results, info
        sent = [e.pixels(total) for e in self.engines]                         # [(N, kp, band_rows, WW), info] per band
        for r, e in enumerate(self.engines):                                   # "all-to-all": rank r receives slice r of every band
            mine = len(range(r, K, N))
            if mine == 0:
                continue
            recv = np.stack([sent[b][0][r] for b in range(N)])                 # (N, kp, band_rows, WW)
            hi, raw, nms, prob = e.hough_stage(recv, kp)
            for j in range(mine):
                i = j * N + r
                info[i] = _merge_info(sent[0][1][i], hi[j])
                results[i] = _package(self.s.num_cls, hi[j], raw[j], nms[j], prob[j])
        return results, info

    def detect_batch_device(self, sub_ptrs, K: int):
        self.submit_device(sub_ptrs, K)
        return self.collect()

    def collect(self, arrays=False):
        """Results of the oldest pass queued with submit_device.  arrays=True (shared Hough context only): (None, info) with the
        pass's frame records as ONE array and no per-frame (lines, cls_pred) objects -- what a batched consumer (the native
        collector, the bench) takes; the raw / NMS segments of the pass stay in `.last_arrays` = (raw, nms, prob) until the next
        collect.  (Packaging 500 frames one by one is 2.5 ms of interpreter time per pass, during which nothing is submitted.)"""
        K, kp = self._queued.pop(0)
        p = self.s.plan
        N = p.nbands
        thr = [e.thresholds_collect() for e in self.engines][0]  # every band frees its slot; the scalars are the same everywhere
        if self.shared and arrays:
            if getattr(self, "_bufs", None) is None or len(self._bufs[0]) < K:
                self._bufs = (np.zeros(K, _lib.FRAME_INFO_DTYPE), np.zeros((K, _lib.MDH_MAX_LINES, 4), np.int32),
                              np.zeros((K, _lib.MDH_MAX_LINES, 4), np.int32), np.zeros((K, _lib.MDH_MAX_LINES), np.float64))
            binfo, raw, nms, prob = self._bufs
            self.hough.collect_into(binfo, raw, nms, prob)
            info = binfo[:K].copy()
            for f in ("bi_threshold", "bi_threshold_float", "snr", "sigma"):
                info[f] = thr[f][:K]
            self.last_arrays = (raw, nms, prob)
            return None, info
        info = np.zeros(K, _lib.FRAME_INFO_DTYPE)
        results = [None] * K
        if self.shared:
            hi, raw, nms, prob = self.hough.hough_planes_collect(K)
            for i in range(K):
                info[i] = _merge_info(thr[i], hi[i])
                results[i] = _package(self.s.num_cls, hi[i], raw[i], nms[i], prob[i])
            return results, info
        for r, e in enumerate(self.engines):
            mine = len(range(r, K, N))
            if mine == 0:
                continue
            hi, raw, nms, prob = e.hough_collect(kp)
            for j in range(mine):
                i = j * N + r
                info[i] = _merge_info(thr[i], hi[j])
                results[i] = _package(self.s.num_cls, hi[j], raw[j], nms[j], prob[j])
        return results, info

    def submit_device(self, sub_ptrs, K: int):
        """Queue a pass (at most two in flight: the exchange buffers alternate); collect() returns the oldest one's results, so
        that the Hough replay of pass p runs under the band stages of pass p + 1.  Everything resident in HBM: sub_ptrs[b] = device
        pointer of band b's rows of the K frames ((K, rows_b, W) u8, as `plan.crop` cuts them).  NOTHING here waits for the GPU
        (round 3): the noise sums are added up on the device (mdh_bands_allreduce_local: what ncclAllReduce does across GPUs), the
        bit planes go band -> send buffer -> receive buffer of the frame's Hough context through queued device-to-device copies
        ordered by events (what the all-to-all moves over xGMI), and the Hough stage is queued behind them."""
        p = self.s.plan
        N = p.nbands
        kp = -(-K // N)
        if not hasattr(self, "_queued"):
            self._queued, self._seq = [], 0
        if len(self._queued) >= 2:
            raise RuntimeError("two passes are already queued: collect() one first")
        if getattr(self, "_xchg_kp", 0) < kp:
            if self._queued:
                raise RuntimeError("the pass size grew while a pass is queued")
            self._free_xchg()
            piece = kp * p.band_rows * p.WW * 8
            ctx = self.engines[0].band
            if self.shared:  # (no exchange buffers: the bands write into the Hough context's planes)
                self._send = [[], []]
                self._recv = [[], []]
            else:
                self._send = [[ctx.device_alloc(N * piece) for _ in range(N)] for _ in range(2)]
                self._recv = [[ctx.device_alloc(N * piece) for _ in range(N)] for _ in range(2)]
            self._xchg_kp = kp
        send, recv = self._send[self._seq % 2], self._recv[self._seq % 2]
        for b, e in enumerate(self.engines):
            e.noise_submit(sub_ptrs[b], K=K, on_device=True)
        _lib.bands_allreduce_local([e.band for e in self.engines])
        if self.shared:
            # band b writes its rows of every plane where the Hough context reads them: that context's own dst planes [K][H][WW]
            import os
            if os.environ.get("MDH_BANDS_INPLACE", "1") == "0":  # A/B: pieces into a buffer of their own, assembled by a copy
                if not recv:
                    piece_b = K * p.band_rows * p.WW * 8
                    self._recv = [[self.engines[0].band.device_alloc(N * piece_b)] for _ in range(2)]
                    recv = self._recv[self._seq % 2]
                piece_b = K * p.band_rows * p.WW * 8
                for b, e in enumerate(self.engines):
                    e.pixels_submit(recv[0] + b * piece_b, nsplit=1)
                    e.band.chain_to(_lib.MDH_STREAM_POST_LAST, self.hough, _lib.MDH_STREAM_POST_NEXT)
                self.hough.hough_planes_submit(recv[0], K, N, p.band_rows, on_device=True)
                self._queued.append((K, kp))
                self._seq += 1
                return
            planes = self.hough.hough_input_planes
            if not planes:
                raise RuntimeError("the shared Hough context has no free pass slot")
            # every band's scalar chain first (one workgroup of sixteen waves each: between two bands' window kernels it would
            # wait for a CU to drain, with the next window kernel behind it), then the pixel stages band by band
            for e in self.engines:
                e.band.band_thresholds_submit(self.s.plan.roi_px)
            # (MDH_BANDS_COUNT_IN_EXPORT=0: rows only, the Hough context counts them in a sweep of its own -- round 3, A/B)
            count = os.environ.get("MDH_BANDS_COUNT_IN_EXPORT", "1") != "0"
            for b, e in enumerate(self.engines):
                if count:
                    d = p.bands[b]
                    e.band.band_pixels_submit_to(p.roi_px, d["own0"] - d["sub0"], d["own1"] - d["own0"], self.hough, d["own0"])
                else:
                    e.pixels_submit_into_planes(planes)
                e.band.chain_to(_lib.MDH_STREAM_POST_LAST, self.hough, _lib.MDH_STREAM_POST_NEXT)
            self.hough.hough_planes_submit(planes, K, 1, p.H, on_device=True)
            self._queued.append((K, kp))
            self._seq += 1
            return
        for b, e in enumerate(self.engines):
            e.pixels_submit(send[b])
        live = kp * p.band_rows * p.WW * 8
        for r, e in enumerate(self.engines):  # "all-to-all": rank r pulls slice r of every band's send buffer
            if not len(range(r, K, N)):
                continue
            for b, eb in enumerate(self.engines):
                eb.band.chain_to(_lib.MDH_STREAM_POST_LAST, e.hough, _lib.MDH_STREAM_POST_NEXT)
                e.hough.device_copy_on(_lib.MDH_STREAM_POST_NEXT, recv[r] + b * live, send[b] + r * live, live)
            e.hough_submit(recv[r], kp, on_device=True)
        self._queued.append((K, kp))
        self._seq += 1

    def _free_xchg(self):
        for ptr in [q for set_ in getattr(self, "_send", []) + getattr(self, "_recv", []) for q in set_]:
            self.engines[0].band.device_free(ptr)
        self._send, self._recv, self._xchg_kp = [], [], 0

    def close(self):
        self._free_xchg()
        if self.hough is not None:
            self.hough.close()
            self.hough = None
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        for e in reversed(self.engines):  # (borrowers of band 0's streams first)
            e.close()


class BandedM3Detector:
    """This rank's share of one stream tiled over `world` GPUs (one process per GPU).

    Same constructor arguments as the reference's detectors (`mask` is the WHOLE frame's mask) plus the rank layout.
    `detect_batch(sub_frames)` takes this rank's rows of K consecutive frames (`plan.crop(frames, rank)`) and returns
    on rank 0 what `M3Detector.detect_batch` returns for the whole frames; other ranks get (None, info)."""

    def __init__(self, window_sec, fps, mask, num_cls, cfg, logger=None, *, rank: int, world: int, group=None, device: int = 0,
                 max_batch: int = 64, engine=HipBandEngine, comm=None):
        """group: this rank's `dist.Group` (or anything with its rank / world / all_reduce_sum_i64 / all_to_all_u64 /
        gather_object -- the CPU tests plug a gloo-backed one in).  comm: an `rccl.Communicator` of this rank (one rank per GPU).
        With it -- or with a group that holds one, from whose control plane a second communicator is made for the library's
        streams -- the data-path collectives are RCCL calls enqueued on the detector library's own streams and a pass is queued
        without a host round trip; the group then only carries the unique id and the final gather of the results.  Without a
        communicator (ranks sharing a device, substitute engines) the blocking steps and host buffers of round 2 are used."""
        self.s = _BandSetup(window_sec, fps, mask, num_cls, cfg, world)
        self.rank, self.world, self.group = int(rank), int(world), group
        if world > 1 and group is None:
            raise ValueError("world > 1 needs a group (metdetpy_amd.dist.init_from_env)")
        self.plan = self.s.plan
        self.engine = engine(self.plan, self.rank, self.s.n, cfg, self.s.mask_area, max_batch, device)
        self.max_batch = int(max_batch)
        self.comm = comm
        self._own_comm = False
        t = float(os.environ.get("MDH_RCCL_TIMEOUT", "120"))
        if self.comm is None and engine is HipBandEngine and getattr(group, "comm", None) is not None:
            from . import rccl
            # (MDH_RCCL_TIMEOUT seconds, default 120: a rank whose peers never arrive raises instead of hanging in ncclCommInitRank)
            self.comm = rccl.Communicator.from_group(group.cp, device=device, timeout_s=t)
            self._own_comm = True
        if self.comm is not None and engine is HipBandEngine:
            # a pass that waits on a dead peer's send must not hang collect(): the library gives up after the same time,
            # collect() then aborts the communicator (which releases the stuck kernels) and raises
            self.engine.band.set_wait_timeout(t)
            self.engine.hough.set_wait_timeout(t)
        self._xbuf = None
        self._pending, self._seq = [], 0

    def _device_exchange_buffers(self):
        """Two send and two receive buffers in HBM, (world, kp_max, band_rows, WW) u64 each, used alternately."""
        if self._xbuf is None:
            p = self.plan
            nbytes = self.world * (-(-self.max_batch // self.world)) * p.band_rows * p.WW * 8
            ctx = self.engine.band
            self._xbuf = [[ctx.device_alloc(nbytes) for _ in range(2)] for _ in range(2)]
        return self._xbuf[0][self._seq % 2], self._xbuf[1][self._seq % 2]

    def _submit_queued(self, sub_frames, K, on_device):
        """RCCL on the library's streams: noise sums all-reduced in place on the noise stream, bit planes exchanged on the post
        stream, the Hough stage chained behind the exchange.  Nothing waits for the GPU."""
        p, N, r = self.plan, self.world, self.rank
        band, hough = self.engine.band, self.engine.hough
        kp = -(-K // N)
        nj = self.engine.noise_submit(sub_frames, K=K, on_device=on_device)
        self.comm.all_reduce_sum_i64(band.band_sums_device, 2 * nj, band.stream_handle(_lib.MDH_STREAM_NOISE))
        send, recv = self._device_exchange_buffers()
        self.engine.pixels_submit(send)
        self.comm.all_to_all_u64(send, recv, kp * p.band_rows * p.WW, band.stream_handle(_lib.MDH_STREAM_POST_LAST))
        mine = len(range(r, K, N))
        if mine:
            band.chain_to(_lib.MDH_STREAM_POST_LAST, hough, _lib.MDH_STREAM_POST_NEXT)
            self.engine.hough_submit(recv, kp, on_device=True)
        self._pending.append((K, kp, mine, None, None, None))
        self._seq += 1

    def submit(self, sub_frames, K=None, on_device=False):
        """Queue a pass: noise sums -> all-reduce -> pixel stages -> all-to-all of the bit planes -> this rank's share of the
        Hough stage, which is only QUEUED (mdh_hough_planes_submit): it runs under the band stages of the next pass.  At most two
        passes may be queued; collect() returns the oldest one's results."""
        p, N, r = self.plan, self.world, self.rank
        K = len(sub_frames) if not on_device else int(K)
        if K > self.max_batch:
            raise ValueError("K exceeds max_batch")
        if len(self._pending) >= 2:
            raise RuntimeError("two passes are already queued: collect() one first")
        if self.comm is not None:
            return self._submit_queued(sub_frames, K, on_device)
        # no communicator: blocking steps, host buffers through the group (ranks sharing a device; substitute engines)
        kp = -(-K // N)
        part = self.engine.noise(sub_frames, K=K, on_device=True) if on_device else self.engine.noise(sub_frames)
        total = part if self.group is None or len(part) == 0 else self.group.all_reduce_sum_i64(part)  # SUM; int64, exact
        sent, thr_info = self.engine.pixels(total)
        # world == 1: the send layout of one band IS the receive layout
        recv = sent if self.group is None else self.group.all_to_all_u64(sent)
        mine = len(range(r, K, N))
        done = None
        if mine:
            if hasattr(self.engine, "hough_submit"):
                self.engine.hough_submit(recv, kp, on_device=False)
            else:  # a substitute engine (CPU tests of the exchange logic) runs its Hough stage at once
                done = self.engine.hough_stage(recv, kp, on_device=False)
        self._pending.append((K, kp, mine, thr_info, done, None))
        self._seq += 1

    def collect(self):
        """Results of the oldest queued pass: on rank 0 what `M3Detector.detect_batch` returns for the whole frames; other ranks
        get (None, info)."""
        N, r = self.world, self.rank
        K, kp, mine, thr_info, done, _keep = self._pending.pop(0)
        try:
            if thr_info is None:  # queued pass: its thresholds are fetched now (this also frees the band context's slot)
                thr_info = self.engine.thresholds_collect()
            if mine and done is None:
                done = self.engine.hough_collect(kp)
        except _lib.MetDetHipError:
            if self.comm is not None:  # timed out (or failed) with collectives queued: release them, then report
                self.comm.abort()
            raise
        part = []
        if mine:
            hi, raw, nms, prob = done
            for j in range(mine):
                part.append((j * N + r, hi[j].copy(), raw[j, :min(int(hi[j]["n_raw"]), _lib.MDH_MAX_LINES)].copy(),
                             nms[j, :int(hi[j]["n_nms"])].copy(), prob[j, :int(hi[j]["n_nms"])].copy()))
        gathered = [part] if self.group is None else self.group.gather_object(part, 0)  # the final detection gather
        if r != 0:
            return None, thr_info
        info = np.zeros(K, _lib.FRAME_INFO_DTYPE)
        results = [None] * K
        for chunk in gathered:
            for i, hi, raw, nms, prob in chunk:
                info[i] = _merge_info(thr_info[i], hi)
                results[i] = _package(self.s.num_cls, hi, raw, nms, prob)
        return results, info

    def detect_batch(self, sub_frames, K=None, on_device=False):
        """sub_frames: this rank's rows of K frames, a (K, rows, W) u8 array -- or a device pointer to them with K given."""
        self.submit(sub_frames, K=K, on_device=on_device)
        return self.collect()

    def close(self):
        if self._xbuf is not None:
            for ptr in [q for pair in self._xbuf for q in pair]:
                self.engine.band.device_free(ptr)
            self._xbuf = None
        if self.comm is not None and self._own_comm:
            self.comm.close()
        self.comm = None
        self.engine.close()
