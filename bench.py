#!/usr/bin/env python3
"""bench.py -- frames/sec of the M3 line-detector hot path on synthetic 1920x1080 uint8 streams (BASELINE config C2).

    python bench.py --gpus N --steps K --warmup W

One "step" = `--passes-per-step` native passes (mdh_submit_batch / mdh_collect_batch, several passes in flight), each over a
batch of `--batch` consecutive frames of a seeded synthetic night-sky stream that is already resident in HBM: update + detect
incl. Hough and NMS, results on the host.  A pass of the headline workload lasts 1.6 ms; SURVEY 8d asks for >= 2 s timed, so a
step is sized (64 passes = 256 000 frames at 1080p, ~0.10 s) such that the driver's `--steps 20` times 2.1 s and the
pipeline's one fill and one drain are < 1 % of the region.  config.frames_per_step / passes_per_step / frames_per_pass say so.
N > 1: one process per GPU (the driver's launcher exports RANK / WORLD_SIZE / MASTER_*), one independent stream per GPU (BASELINE
config C4), no data-path collective; the collectives are the barrier around the timed region (rendezvous store), the
max-over-ranks time and the final gather of the detection counts (RCCL through ctypes: no torch in this script or the package).  `value` = frames all ranks processed / max-over-ranks time.

The JSON line also carries (rank 0):
  roofline      the dominant pixel kernel (stack_threshold), priced PHYSICALLY: the bytes one launch must move (every frame
                byte once + the window carried into the launch + what it stores) over (a) the time the timed region spent per
                pass -- `frac`, tied to ms_per_step, the number the driver's own clock checks -- and (b) the launch's own
                duration measured live with HIP events on the library's stream with one pass in flight and nothing overlapping
                (`frac_isolated`; rocprofv3 --kernel-trace of `--depth 1` averages the same launches).  SURVEY 8d's
                "(n+1) reads per pixel" model is kept as `algorithmic_frac`: the kernel is time-blocked (each byte is fetched
                once per launch, not n times), so that figure exceeds 1 by design and is not a bandwidth.
  cpu_baseline  the CPU oracle timed on this box's host cores over a bounded sample of the same stream (N = 1 only): the
                C port (oracle/metdet_oracle.c) at 1 thread and at the OpenMP team sizes tried, and the structure-faithful
                NumPy oracle on one thread.
  parity        GPU vs oracle on the first frames of the stream, down to the collector's records.

`secondary` (N = 1, headline run only): the other workloads of BASELINE.json and the per-frame protocol, each run by this same
script in a subprocess right after the headline region (>= 1 s timed each, parity on >= 64 frames each): C3 (4K, n = 16, mask),
C5 (8K, m3det_high), C1s (960x540, the reference's default runtime size), per-frame from pinned / recycled pageable / fresh
pageable frames.  --no-secondary skips them.

Other timed entries:  --workload C3|C5|C1s|C1e,  --protocol per-frame (the reference's update()/detect() loop from host
frames),  --mode bands|timeshard (ONE stream tiled into row bands / cut into time windows; N bands or windows on one
device when WORLD_SIZE is 1),  --host-input (frames cross PCIe every pass).
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (GPU_MAX_HW_QUEUES: not set here -- the binding asks for what its mode wants before the HIP runtime loads, metdetpy_amd/_lib.py)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s is what a streaming copy reaches)

WORKLOADS = {
    "C2": dict(H=1080, W=1920, n=8, seed=20220413, batch=4000, pps=64, desc="synthetic 1920x1080 u8 stream, 8-frame window"),
    "C3": dict(H=2160, W=3840, n=16, seed=20220414, batch=1000, pps=64, desc="synthetic 3840x2160 u8 stream, 16-frame window + mask"),
    "C1s": dict(H=540, W=960, n=8, seed=20220412, batch=16000, pps=48, desc="synthetic 960x540 u8 stream (runtime size of config 1)"),
    "C5": dict(H=4320, W=7680, n=8, seed=20220415, batch=500, pps=24, sensitivity="high",
               desc="synthetic 7680x4320 u8 stream, m3det_high thresholds, whole frame on one GPU (no row-band tiling)"),
    "C1e": dict(H=540, W=960, n=8, seed=20220412, batch=500, src=(2160, 3840),
                desc="config-1 substitute: synthetic 3840x2160x3 BGR clip in HBM -> resize 960x540 + gray + mask-east (loader "
                     "restatement on the GPU) -> detector -> native collector"),
}
STREAK_PERIOD = 50  # SkyStream default: one streak per 50 frames
# --stream: SkyStream arguments, the stream's period (the resident pool holds whole periods) and the detector preset it runs under
STREAMS = {
    "sky": dict(kw={}, period=50),
    "noisy": dict(kw=dict(sigma=6.0), period=50, sensitivity="low"),
    "hot": dict(kw=dict(hot_frac=0.01), period=50),
    "cloud": dict(kw=dict(cloud_period=50), period=50),
    "sparse": dict(kw=dict(streak_period=400), period=400),
    "dense": dict(kw=dict(streak_period=24), period=24),
}
STREAM = "sky"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20: >= 2 s with the default passes per step, SURVEY 8d)")
    ap.add_argument("--warmup", type=int, default=2, help="untimed steps in front")
    ap.add_argument("--passes-per-step", type=int, default=None, help="native passes per step (default per workload: 64 at 1080p and 4K, "
                    "48 at 960x540, 24 at 8K -- a step of ~0.1 s; the per-frame protocol and --mode bands|timeshard keep 1)")
    ap.add_argument("--batch", type=int, default=None, help="frames per PASS (and frames resident in HBM per GPU)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` block (the other workloads and the per-frame protocol, "
                    "run as subprocesses after the headline region when this is the plain headline command on one GPU)")
    ap.add_argument("--frames", default="pinned", choices=["pinned", "recycled", "pageable"], help="--protocol per-frame: where the loader's "
                    "frames live -- a pinned pool of mdh_host_alloc; a ring of 8 ordinary NumPy buffers refilled in turn (ProcessVideoLoader's "
                    "RawArray ring: the library registers them on their third hand-over); a fresh NumPy array per frame (ThreadVideoLoader)")
    ap.add_argument("--unique", type=int, default=250, help="period of the synthetic stream in frames, a multiple of the streak "
                    "period: the resident batch is frames[i %% unique], so no streak is cut where the stream wraps (0 = every "
                    "resident frame distinct; generation is NumPy on the host and dominates the start-up time).  The default "
                    "passes (4000 / 2000 / 1000 / 500 frames) are whole multiples of it, so a resubmitted pass CONTINUES the "
                    "stream (frame 249 -> frame 0 is seamless) instead of starting on a scene cut")
    ap.add_argument("--workload", default="C2", choices=sorted(WORKLOADS))
    ap.add_argument("--window", type=int, default=None, help="override the workload's window length n (the reference's default `window_sec: 1` "
                    "gives n = 25 / 30 at 25 / 30 frames per second, Detector.py:200; BASELINE's configs quote n = 8 and 16)")
    ap.add_argument("--stream", default="sky", choices=sorted(STREAMS), help="the synthetic stream's weather: sky = SURVEY 8d's friendly one "
                    "(N(30, 2.5), one streak per 50 frames); noisy = sigma 6 under m3det_low (Hough 10 / 15 / 8); hot = 1 %% hot pixels; cloud = a "
                    "cloud-edge frame every 50 frames (> 8192 lit pixels for n frames); sparse / dense = a streak in ~5 %% / ~80 %% of the frames")
    ap.add_argument("--mode", default="streams", choices=["streams", "bands", "timeshard"])
    ap.add_argument("--protocol", default="batch", choices=["batch", "per-frame"])
    ap.add_argument("--detector", default="m3", choices=["m3", "classic"], help="classic: the reference's second line detector "
                    "(ClassicDetector, Detector.py:245-296: 4-frame ring, fixed threshold 20 here), same stream and harness")
    ap.add_argument("--shards", type=int, default=8, help="--mode bands|timeshard with one process: bands / windows on this device")
    ap.add_argument("--cpu-seconds", type=float, default=16.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--parity-frames", type=int, default=256, help="frames of the stream compared against the CPU oracle "
                    "after the timed region (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--standalone-passes", type=int, default=12, help="passes run one at a time after the timed region to "
                    "measure the non-overlapped duration of every kernel (0 = skip)")
    ap.add_argument("--hough-mode", default="exact", choices=["exact", "accumulator"], help="accumulator: the order-free "
                    "full-accumulator transform (MDH_HOUGH_ACCUMULATOR) instead of the exact replay of cv2.HoughLinesP; parity is then "
                    "the tolerance rule of SURVEY 8d (collector records matched against the exact mode's), not bit-exactness")
    ap.add_argument("--sensitivity", default=None, choices=["low", "normal", "high"], help="override the workload's detector cfg "
                    "(experiments: C5's frame size at m3det_normal thresholds)")
    ap.add_argument("--depth", type=int, default=0, help="passes in flight in the timed region (0 = as many as the library "
                    "allows; 1 = one at a time: nothing overlaps, the kernel spans are the kernels' own durations)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live HBM-traffic leg (roofline.traffic): two short re-executions of this "
                    "command under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `WRITE_SIZE` (separate passes) when rocprofv3 is on PATH")
    ap.add_argument("--with-collector", action="store_true", help="the native MeteorCollector inside the timed region: every collected pass is "
                    "handed to mdh_collector_update_batch on a second host thread (collector.py:420-528); `value` then counts frames whose "
                    "results have gone through the collector, and the line carries a `collector` object (records, seconds of the thread's work)")
    ap.add_argument("--host-input", action="store_true", help="frames come from PINNED HOST memory every pass (PCIe-inclusive "
                    "rate; never the headline value, see DESIGN.md)")
    return ap.parse_args()


SENSITIVITY = "normal"


DETECTOR = "m3"
HOUGH_MODE = "exact"
CLASSIC_THRESHOLD = 20  # |x_t - x_(t-1)| of the stream's N(30, 2.5) background has sigma 3.5: 20 keeps the noise out, a +60 streak in


def make_cfg():
    from metdetpy_amd import BinaryCfg, BinaryCoreCfg, DynamicCfg, HoughLineCfg
    if DETECTOR == "classic":  # fixed threshold (the oracle's ClassicOracle restates the detect() half of the class)
        return BinaryCfg(BinaryCoreCfg(False, CLASSIC_THRESHOLD, SENSITIVITY, 0.1, 2), HoughLineCfg(10, 10, 10), DynamicCfg(True, 5))
    # config/m3det_normal.json (or m3det_high.json / m3det_low.json: the low preset's Hough block is 10 / 15 / 8) detector block
    hough = HoughLineCfg(10, 15, 8) if SENSITIVITY == "low" else HoughLineCfg(10, 10, 10)
    return BinaryCfg(BinaryCoreCfg(True, 7, SENSITIVITY, 0.1, 2), hough, DynamicCfg(True, 5))


def oracle_detector(n, mask):
    """The C port of the oracle under the same preset as make_cfg()."""
    from metdetpy_amd.utils import select_std_roi
    from oracle import c_oracle as co
    c = make_cfg()
    return co.CM3(n, mask, select_std_roi(mask, 0.1), sensitivity=SENSITIVITY, hough_threshold=c.hough_line.threshold,
                  hough_min_len=c.hough_line.min_len, hough_max_gap=c.hough_line.max_gap)


def make_detector(n, mask, device, max_batch):
    from metdetpy_amd import ClassicDetector, M3Detector
    if DETECTOR == "classic":
        return ClassicDetector(1, 4.0, mask, 10, make_cfg(), None, device=device, max_batch=max_batch)
    return M3Detector(1, n, mask, 10, make_cfg(), None, device=device, max_batch=max_batch, hough_mode=HOUGH_MODE)


def parity_check_accumulator(frames, n, mask, local, count):
    """hough_mode = accumulator is a different algorithm from cv2.HoughLinesP: its parity is SURVEY 8d's tolerance rule.  The
    same frames go through the exact-mode detector (itself bit-exact against the oracle, see the default run) and the
    accumulator-mode one, both through the native collector; every record of either must have a partner in the other with
    time IoU >= 0.3 and box IoU >= 0.3 (evaluate.py's matching rule)."""
    from metdetpy_amd import M3Detector
    from metdetpy_amd.collector import NativeMeteorCollector
    from metdetpy_amd.metstruct import CollectorCfg, MeteorCfg, RecheckSwitch, RuntimeParams
    from metdetpy_amd.pipeline import calculate_area_iou, calculate_time_iou
    count = min(count, len(frames))
    H, W = mask.shape
    ccfg = CollectorCfg(MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})
    rt = RuntimeParams(fps=float(n), exp_frame=1, eq_fps=float(n), eq_int_fps=n, exp_time=1.0 / n, runtime_size=[W, H],
                       raw_size=[W, H], positive_category_list=["METEOR", "RED_SPRITE"])

    class Box:
        def __init__(self, t):
            (self.x1, self.x2), (self.y1, self.y2) = sorted([t.pt1[0], t.pt2[0]]), sorted([t.pt1[1], t.pt2[1]])

    recs, segs = {}, {}
    for mode in ("exact", "accumulator"):
        det = M3Detector(1, n, mask, 10, make_cfg(), None, device=local, max_batch=count, hough_mode=mode)
        info, _raw, nms, prob = det._ctx.process_batch(np.ascontiguousarray(frames[:count]))
        det.close()
        col = NativeMeteorCollector(ccfg, rt)
        col.update_batch(0, 0, info, nms, prob)
        col.clear()
        recs[mode] = [t for r in col.met_exporter.meteor_list for t in r.target]
        segs[mode] = int(info["n_raw"].sum())

    def unmatched(a_list, b_list):
        return sum(1 for a in a_list if not any(calculate_time_iou(a, b) >= 0.3 and calculate_area_iou(Box(a), Box(b)) >= 0.3 for b in b_list))

    miss = unmatched(recs["accumulator"], recs["exact"]) + unmatched(recs["exact"], recs["accumulator"])
    return {"frames_checked": count, "records_exact": len(recs["exact"]), "records_accumulator": len(recs["accumulator"]),
            "unmatched_records": miss, "raw_segments_exact": segs["exact"], "raw_segments_accumulator": segs["accumulator"],
            "verdict": "matched under tolerance" if miss == 0 and recs["exact"] else "MISMATCH",
            "checked": "collector records of the accumulator-mode detector vs the exact-mode detector, time IoU >= 0.3 and box IoU >= 0.3 both ways"}


def parity_check_classic(frames, mask, local, count):
    """ClassicDetector vs the NumPy oracle (PPHT from the C port) on the first frames of the stream: lit counts, raw segments."""
    from oracle import c_oracle as co
    from oracle import m3_oracle as mo
    count = min(count, len(frames), 48)
    det = make_detector(4, mask, local, count)
    info, raw, _nms, _prob = det._ctx.process_batch(np.ascontiguousarray(frames[:count]))
    det.close()
    orc = mo.ClassicOracle(mask, threshold=CLASSIC_THRESHOLD, hough_threshold=10, hough_min_len=10, hough_max_gap=10, hough_fn=co.hough_lines_p)
    bad, segs = [], 0
    for i in range(count):
        orc.update(frames[i])
        o_lines, _ = orc.detect()
        lit = int(np.count_nonzero(orc.dst)) if (orc.dst is not None and i >= 3) else 0
        ok = int(info[i]["n_raw"]) == len(o_lines) and int(info[i]["lit_count"]) == lit and \
            np.array_equal(raw[i][:min(len(o_lines), 512)], np.asarray(o_lines).reshape(-1, 4)[:512])
        segs += len(o_lines)
        if not ok:
            bad.append(i)
    return {"frames_checked": count, "raw_segments_checked": segs, "mismatching_frames": bad,
            "verdict": "identical" if not bad else "MISMATCH",
            "checked": "lit_count, raw Hough segments (order included) vs oracle/m3_oracle.py ClassicOracle"}


def load_mask(H, W):
    """test/mask-east.jpg binarised by the loader rule (fixture committed as tests/golden/mask_east_*.npz)."""
    path = os.path.join(ROOT, "tests", "golden", f"mask_east_{W}x{H}.npz")
    if os.path.exists(path):
        return np.load(path)["mask"]
    return None


def _time_detector(step, frames, warm, seconds, cap):
    for i in range(warm):
        step(frames[i])
    done, t0 = 0, time.perf_counter()
    while True:
        step(frames[(warm + done) % len(frames)])
        done += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or done >= cap:
            return done, dt


def cpu_baseline(frames, n, mask, seconds):
    """Time the CPU oracle (update + detect up to raw Hough + NMS) on the host cores: the C port at 1 thread and at larger
    OpenMP teams (the best is `value`), and the NumPy oracle -- structure-faithful to the reference: ring + np.max, second
    ring for the dynamic mask -- on one thread."""
    from oracle import c_oracle as co
    from oracle import m3_oracle as mo
    from metdetpy_amd.utils import select_std_roi
    roi = select_std_roi(mask, 0.1)
    avail = len(os.sched_getaffinity(0))
    warm = min(2 * n, len(frames))
    teams = sorted({1, min(16, avail), min(64, avail)})
    share = seconds / (len(teams) + 1.5)
    by_threads = {}
    for th in teams:
        co.lib().om3_set_threads(th)
        det = oracle_detector(n, mask)

        def step(f, det=det):
            det.update(f)
            raw = det.detect()
            if 0 < len(raw) <= 500:
                co.lineset_nms(raw)

        done, dt = _time_detector(step, frames, warm, share, 4 * len(frames))
        by_threads[int(co.lib().om3_num_threads())] = dict(value=done / dt, frames=done)
    c = make_cfg()
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, sensitivity=SENSITIVITY, hough_threshold=c.hough_line.threshold,
                      hough_min_len=c.hough_line.min_len, hough_max_gap=c.hough_line.max_gap)

    def np_step(f):
        orc.update(f)
        orc.detect()

    np_done, np_dt = _time_detector(np_step, frames, min(warm, n), 1.5 * share, 4 * len(frames))
    best = max(by_threads, key=lambda k: by_threads[k]["value"])
    return dict(value=by_threads[best]["value"], unit="frames/s", cores=best, kind="port",
                sample=f"{by_threads[best]['frames']} frames of the same stream after {warm} warm-up frames, oracle/metdet_oracle.c "
                       f"(gcc -O3 -fopenmp, {best} threads: the best of the teams tried), host cpu_count={os.cpu_count()}",
                c_port_by_threads={str(k): v["value"] for k, v in by_threads.items()},
                single_thread={"c_port": by_threads[min(by_threads)]["value"], "numpy_oracle": np_done / np_dt,
                               "numpy_oracle_sample": f"{np_done} frames, oracle/m3_oracle.py (NumPy stages as the reference's, "
                                                      "OpenCV-side ops and PPHT from the C port)"})


def parity_check(frames, n, mask, local, count, timed=None):
    """Detection parity vs the CPU reference restatement (the metric's second half): a fresh GPU detector and the C
    port of the oracle see the first `count` frames of the stream; thresholds, lit counts, raw Hough segments (in
    discovery order) and NMS output must be identical, and so must the records the collector makes of them.

    timed = {"tally": [raw segments, NMS segments, lit pixels] summed over the TIMED region, "frames": its frame count,
    "per_pass_equal": every timed pass gave the same tallies}: the timed region processes the stream's period over and over in
    its steady state (the threshold has settled, every window is full), so its tallies must be the oracle's tallies of one
    steady-state period times the number of periods -- checked here against the LAST whole period the oracle saw (count is
    raised to at least two whole periods -- the first carries the warm-up of the windows; see
    below for how the oracle gets from there to the timed region)."""
    from metdetpy_amd import M3Detector
    from metdetpy_amd.utils import select_std_roi
    from oracle import c_oracle as co
    period = len(frames)
    # (the resident pool must hold whole periods: a pass of 125 frames over a 50-frame period is resubmitted on a scene cut, and the
    # stream the detector sees is then not the periodic one the oracle's chain assumes)
    want_steady = timed is not None and timed["frames"] % period == 0 and timed.get("pool_frames", period) % period == 0 and count > 0
    if want_steady:
        # whole periods: three where a frame is cheap for the oracle (the threshold has usually settled by the third, which then
        # serves as the steady period itself), two at 4K / 8K (the second one starts with full windows)
        count = max(count, (3 if mask.size <= 1080 * 1920 else 2) * period) // period * period
        want_steady = timed["warm_frames"] >= count  # (the timed region must lie beyond what the oracle runs through frame by frame)
    src = frames
    co.lib().om3_set_threads(min(16, len(os.sched_getaffinity(0))))
    orc = oracle_detector(n, mask)
    o_rows, o_results, o_lines_num, o_lit = [], [], [], []

    def period_tally(k):  # oracle tallies of period k
        lo = k * period
        return [sum(o_lines_num[lo:lo + period]), sum(len(o_results[i][0]) for i in range(lo, lo + period)), sum(o_lit[lo:lo + period])]

    i = 0
    while i < count:
        orc.update(src[i % period])
        o_raw = orc.detect()
        o_rows.append((orc.state()[0], o_raw))
        o_results.append((np.zeros((0, 4), np.int32), np.zeros(0)))
        o_lines_num.append(int(orc.lines_num))
        o_lit.append(int(orc.lit_count))
        if 0 < orc.lines_num <= 500:
            o_results[i] = co.lineset_nms(o_raw)
        i += 1
    steady = None
    if want_steady:
        # Where does the threshold stand in the TIMED region, millions of frames into the stream?  Its EMA keeps creeping for
        # thousands of frames (4K, n = 16: T = 10 from frame 500 to frame 1500, 11 later), so the oracle's third period is not
        # "the steady state".  The stream is periodic, so the oracle's own scalar chain can be carried on without pixels: the
        # sigma of a due frame depends on t mod period only (oracle/m3_oracle.py: threshold_far_into_a_periodic_stream, bit-identical
        # to running the detector through every frame -- tests/test_oracle_golden.py).  The pixel oracle then runs two more periods
        # at that threshold (fixed-threshold mode: the second one is tallied) unless its last period was already there.
        from oracle import m3_oracle as mo
        c_ = make_cfg()
        t_first, t_last = timed["warm_frames"] + 1, timed["warm_frames"] + timed["frames"]
        thr_set, snr_end = mo.threshold_far_into_a_periodic_stream(src, n, select_std_roi(mask, 0.1), c_.binary.interval, SENSITIVITY, count,
                                                                   orc.state()[2], t_first, t_last)
        steady = {"thresholds_in_timed_region": thr_set, "oracle_snr_at_its_end": snr_end}
        last = range(count - period, count)
        if len(thr_set) == 1 and all(o_rows[i][0] == thr_set[0] for i in last):
            steady["tally"] = period_tally(count // period - 1)
            steady["from"] = "the oracle's last whole period (already at that threshold)"
        elif len(thr_set) == 1:
            fx = co.CM3(n, mask, select_std_roi(mask, 0.1), adaptive=False, init_value=thr_set[0], hough_threshold=c_.hough_line.threshold,
                        hough_min_len=c_.hough_line.min_len, hough_max_gap=c_.hough_line.max_gap)
            tl = [0, 0, 0]
            for i2 in range(2 * period):
                fx.update(src[i2 % period])
                r2 = fx.detect()
                if i2 >= period:
                    tl[0] += int(fx.lines_num)
                    tl[2] += int(fx.lit_count)
                    if 0 < fx.lines_num <= 500:
                        tl[1] += len(co.lineset_nms(r2)[0])
            steady["tally"] = tl
            steady["from"] = f"a second oracle detector held at T = {thr_set[0]} (fixed-threshold mode), its second period"
    # the parity detector also records what every frame's Hough replay cost (MDH_HOUGH_DEBUG: cycle stamps per frame of its one
    # pass -- the instrumented instantiations are slower, which is why the timed detector never runs them)
    had_dbg = os.environ.get("MDH_HOUGH_DEBUG")
    os.environ["MDH_HOUGH_DEBUG"] = "1"
    det = M3Detector(1, n, mask, 10, make_cfg(), None, device=local, max_batch=count)
    if had_dbg is None:
        del os.environ["MDH_HOUGH_DEBUG"]
    info, raw, nms, prob = det._ctx.process_batch(np.ascontiguousarray(src[np.arange(count) % period]))
    hough_cost = None
    try:
        import ctypes as C
        dbg2 = np.zeros((2, count, 8), np.int64)
        if det._ctx.L.mdh_debug_hough(C.c_void_p(det._ctx.handle), dbg2.ctypes.data_as(C.POINTER(C.c_longlong)), count) == 0:
            pts, tier, us = dbg2[0][:, 0], dbg2[0][:, 1] & 0xFF, dbg2[0][:, 3] / 100.0
            act = pts > 0
            if act.any():
                w = int(np.argmax(np.where(act, us, -1)))
                hough_cost = {"frames_with_points": int(act.sum()), "replay_us_median": float(np.median(us[act])), "replay_us_p99": float(np.percentile(us[act], 99)),
                              "slowest_frame": {"frame": w, "replay_us": float(us[w]), "points": int(pts[w]), "tier": int(tier[w]), "segments": int(info[w]["n_raw"])},
                              "frames_by_tier": {str(int(k)): int((tier[act] == k).sum()) for k in np.unique(tier[act])},
                              "max_points": int(pts.max()), "note": "per-frame replay time of the Hough stage from the kernels' own cycle stamps (MDH_HOUGH_DEBUG, "
                              "100 MHz wall clock), over the parity leg's frames; tiers: 0 small LDS, 1 large LDS, 2 hybrid, 3 one wave (> 8192 points), 4 tiny"}
    except Exception as e:  # noqa: BLE001
        hough_cost = {"error": f"{type(e).__name__}: {e}"[:200]}
    det.close()
    bad, segs = [], 0
    for i in range(count):
        thr, o_raw = o_rows[i]
        ok = (int(info[i]["bi_threshold"]) == thr and int(info[i]["lit_count"]) == o_lit[i]
              and int(info[i]["n_raw"]) == o_lines_num[i]
              and np.array_equal(raw[i][:min(o_lines_num[i], 512)], o_raw[:512]))
        if 0 < o_lines_num[i] <= 500:
            o_nms, o_prob = o_results[i]
            k = int(info[i]["n_nms"])
            ok = ok and k == len(o_nms) and np.array_equal(nms[i][:k], o_nms) and np.array_equal(prob[i][:k], o_prob)
        segs += o_lines_num[i]
        if not ok:
            bad.append(i)
    # the records a collector makes of them (config/m3det_normal.json's meteor_cfg, re-check off): GPU results through the
    # native collector one pass at a time, oracle results through the oracle-side Python collector frame by frame
    from metdetpy_amd.collector import NativeMeteorCollector
    from metdetpy_amd.metstruct import CollectorCfg, MeteorCfg, RecheckSwitch, RuntimeParams
    from oracle.collector_oracle import MeteorCollector
    H, W = mask.shape
    ccfg = CollectorCfg(MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})
    rt = RuntimeParams(fps=float(n), exp_frame=1, eq_fps=float(n), eq_int_fps=n, exp_time=1.0 / n, runtime_size=[W, H],
                       raw_size=[W, H], positive_category_list=["METEOR", "RED_SPRITE"])
    gcol, ocol = NativeMeteorCollector(ccfg, rt), MeteorCollector(ccfg, rt)
    gcol.update_batch(0, 0, info, nms, prob)
    gcol.clear()
    for i in range(count):
        k = len(o_results[i][0])
        if k or i % rt.eq_int_fps == 0:
            cates = np.zeros((k, 10))
            cates[:, -1] = o_results[i][1]
            cates[:, 0] = 1 - o_results[i][1]
            ocol.update(i, o_results[i][0], cates)
    ocol.clear()
    g_rec, o_rec = [r.to_dict() for r in gcol.met_exporter.meteor_list], [r.to_dict() for r in ocol.met_exporter.meteor_list]
    rec_ok = g_rec == o_rec
    out = {"frames_checked": count, "raw_segments_checked": segs, "mismatching_frames": bad, "records": len(o_rec),
           "records_identical": rec_ok, "verdict": "identical" if not bad and rec_ok else "MISMATCH",
           "checked": "bi_threshold, lit_count, raw Hough segments (order included), lineset_nms output, collector records "
                      "vs oracle/metdet_oracle.c"}
    if hough_cost is not None:
        out["hough_frame_cost"] = hough_cost
    if timed is not None:
        if steady is None:
            out["timed_region"] = {"verdict": "not checked", "why": "the resident pool (or the timed region) is not a whole number of stream periods, or the "
                                   "warm-up is shorter than the oracle's frame-by-frame leg"}
        elif "tally" not in steady:
            out["timed_region"] = {"verdict": "not checked", "why": "the oracle's threshold changes inside the timed region", **steady}
        else:
            periods = timed["frames"] // period
            want = [v * periods for v in steady["tally"]]
            ok = want == list(timed["tally"]) and timed.get("per_pass_equal", True)
            out["timed_region"] = {"periods": periods, "frames": timed["frames"], "first_frame": timed["warm_frames"] + 1,
                                   "oracle_threshold": steady["thresholds_in_timed_region"][0], "oracle_tally_per_period": steady["tally"],
                                   "oracle_tally_from": steady["from"], "expected_tally": want, "gpu_tally": list(timed["tally"]),
                                   "gpu_every_pass_equal": bool(timed.get("per_pass_equal", True)),
                                   "verdict": "identical" if ok else "MISMATCH",
                                   "checked": "[raw segments, NMS segments, lit pixels] summed over every frame of the timed region == the "
                                              "oracle's tallies of one period at the threshold ITS OWN scalar chain reaches there x the number of periods"}
            if not ok:
                out["verdict"] = "MISMATCH"
    return out


def parity_check_per_frame(frames, n, mask, local, count, kind):
    """The per-frame protocol against the C port of the oracle: a fresh detector is fed the first `count` frames of the stream one
    update() / detect() at a time from the same kind of host buffers as the timed loop (pinned pool / recycled ring / fresh arrays);
    thresholds, lit counts, raw Hough segments (order included) and the NMS output must be identical."""
    import ctypes
    from metdetpy_amd import M3Detector
    from metdetpy_amd.utils import select_std_roi
    from oracle import c_oracle as co
    count = min(count, len(frames))
    H, W = mask.shape
    det = M3Detector(1, n, mask, 10, make_cfg(), None, device=local, max_batch=8)
    pool = None
    if kind == "pinned":
        src = np.ascontiguousarray(frames[:count])
        pool = det._ctx.host_alloc(src.nbytes)
        ctypes.memmove(pool, src.ctypes.data, src.nbytes)
        view = np.ctypeslib.as_array((ctypes.c_uint8 * src.nbytes).from_address(pool)).reshape(src.shape)
        source = lambda k: view[k]
    elif kind == "recycled":
        ring = [np.empty((H, W), np.uint8) for _ in range(8)]
        det._ctx.set_auto_register(True)  # the loader's opt-in (it owns a fixed ring and outlives the detector)

        def source(k):
            np.copyto(ring[k % 8], frames[k])
            return ring[k % 8]
    else:
        source = lambda k: frames[k].copy()
    co.lib().om3_set_threads(min(16, len(os.sched_getaffinity(0))))
    orc = oracle_detector(n, mask)
    bad, segs = [], 0
    for i in range(count):
        det.update(source(i))
        lines, _pred = det.detect()
        orc.update(frames[i])
        o_raw = orc.detect()
        ok = det.bi_threshold == orc.state()[0] and int(det._ctx._info1[0]["lit_count"]) == orc.lit_count and det.lines_num == orc.lines_num
        if ok and 0 < orc.lines_num <= 500:
            o_nms, _ = co.lineset_nms(o_raw)
            ok = np.array_equal(det.linesp_ext.reshape(-1, 4)[:512], o_raw[:512]) and np.array_equal(np.asarray(lines).reshape(-1, 4), o_nms)
        segs += orc.lines_num
        if not ok:
            bad.append(i)
    stats = det._ctx.update_stats()
    if pool is not None:
        del view
        det._ctx.host_free(pool)
    det.close()
    return {"frames_checked": count, "raw_segments_checked": segs, "mismatching_frames": bad, "update_stats": stats,
            "verdict": "identical" if not bad else "MISMATCH",
            "checked": f"update() / detect() per frame from {kind} host frames: bi_threshold, lit_count, raw Hough segments (order included), "
                       "lineset_nms output vs oracle/metdet_oracle.c"}


class Dist:
    """The three collectives of the timed harness, without torch: `metdetpy_amd.dist.Group` = the rendezvous store for the
    control plane (the launcher's RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT; barrier, ncclUniqueId) + RCCL through ctypes for
    the data (max-over-ranks time: ncclAllReduce(max); the final detection gather: ncclAllGather, on a private HIP stream).  RCCL
    refuses two ranks on one device: on a box with fewer GPUs than ranks the ranks share devices round-robin and those few bytes
    go through the store instead (`config.collectives` says which; MDH_DIST_BACKEND=store forces it)."""

    def __init__(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.g = None
        # MDH_BENCH_FORCE_DIST=1: join a job of ONE rank too (exercises the RCCL branch -- communicator, all-reduce, all-gather
        # on device memory -- on a one-GPU box)
        if self.world > 1 or os.environ.get("MDH_BENCH_FORCE_DIST") == "1":
            from metdetpy_amd import dist
            self.g = dist.Group(self.rank, self.local, self.world)
            if self.g.backend == "store" and self.g.rccl_note is not None and os.environ.get("MDH_DIST_BACKEND") != "store":
                # Every rank has a GPU of its own and RCCL still could not be used: the run FAILS -- the
                # store is the fall-back for ranks that share a device (the one-GPU dry runs), not a way for a broken RCCL path to
                # look like a pass.  (every rank takes this branch: the outcomes were exchanged; MDH_DIST_BACKEND=store asks for the store)
                if self.rank == 0:
                    print(f"bench.py: {self.world} ranks with a GPU each, but RCCL is not usable ({self.g.rccl_note}); refusing to fall back "
                          "to the rendezvous store (set MDH_DIST_BACKEND=store to run the collectives through it on purpose)", file=sys.stderr, flush=True)
                self.g.close()
                sys.exit(3)
            self.local = self.g.device
            if os.environ.get("MDH_BENCH_VERBOSE_RANKS") == "1":  # tools/run_8gpu.sh: who sits where
                print(f"[rank {self.rank}] of {self.world}: {json.dumps(self.g.describe())}", file=sys.stderr, flush=True)

    @property
    def backend(self):
        return self.g.backend if self.g is not None else None

    def fence(self):
        # the library has no work in flight between run() calls (every pass is collected); the barrier aligns the ranks
        if self.g is not None:
            self.g.barrier()

    def max_time_and_sum(self, dt, counts):
        self.per_rank = [list(counts)]
        if self.g is None:
            return dt, list(counts)
        tmax = self.g.max_f64(dt)
        self.per_rank = self.g.all_gather_i64(counts)  # the final detection gather (RCCL when every rank has its own GPU)
        return tmax, [sum(col) for col in zip(*self.per_rank)]

    def all_gather_object(self, obj):
        return [obj] if self.g is None else self.g.all_gather_object(obj)

    def gather_object(self, obj):
        return [obj] if self.g is None else self.g.gather_object(obj, 0)

    def close(self):
        if self.g is not None:
            self.g.close()


def make_stream_frames(wl, rank, mask, B, unique):
    """The frames one GPU keeps resident: a seeded continuous stream; with unique < B its first `unique` frames, a whole
    number of streak periods, so that the stream is seamless where the resident batch repeats them."""
    from metdetpy_amd.synth import SkyStream
    H, W, n = wl["H"], wl["W"], wl["n"]
    st = STREAMS[STREAM]
    stream = SkyStream(H, W, wl["seed"] + rank, mask01=None if mask.all() else mask, **st["kw"])
    uniq = B if unique <= 0 else min(B, max(unique, 2 * n))
    if uniq < B:
        uniq = max(st["period"], uniq // st["period"] * st["period"])
    return stream.take(uniq)


def pmc_traffic_live(a, dense, kern=None):
    """roofline.traffic measured IN THIS RUN: the same command re-executed for 4 passes (one at a time) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE`, then `--pmc WRITE_SIZE` -- separate passes, as MI355X_MICROARCH.md's HBM section
    prescribes -- and the dominant kernel's counter summed per dispatch, averaged over its launches.  FETCH_SIZE (KB) is doubled:
    gfx950 tallies the 128-byte requests of wide coalesced reads at 64 bytes (same section); WRITE_SIZE is taken as reported.
    Returns (bytes per launch or None, source text)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 is not on PATH"
    kern = kern or ("k_classic_diff" if DETECTOR == "classic" else "k_stack_threshold")
    base = [sys.executable, os.path.abspath(__file__), "--workload", a.workload] + (["--window", str(a.window)] if a.window is not None else []) + ["--stream", a.stream, "--batch", str(a.batch), "--unique", str(a.unique),
            "--steps", "3", "--warmup", "1", "--passes-per-step", "1", "--no-secondary", "--depth", "1", "--cpu-seconds", "0", "--parity-frames", "0", "--standalone-passes", "0",
            "--no-profile", "--no-pmc", "--detector", a.detector, "--hough-mode", a.hough_mode]
    if a.sensitivity:
        base += ["--sensitivity", a.sensitivity]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if dense:
        env["MDH_DENSE"] = "1"  # the timed run ended in the dense output format: measure that one (4 passes would not get there)
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "r", "--output-format", "csv", "--"] + base
            try:
                subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900, check=True)
            except Exception as e:  # noqa: BLE001
                return None, f"rocprofv3 --pmc {counter} failed: {type(e).__name__}"
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                return None, f"rocprofv3 --pmc {counter} wrote no counter_collection.csv"
            total, launches = 0.0, set()
            for row in csv.DictReader(open(files[0])):
                if row.get("Counter_Name") == counter and kern in row.get("Kernel_Name", ""):
                    total += float(row["Counter_Value"])
                    launches.add(row["Dispatch_Id"])
            if not launches:
                return None, f"no {kern} dispatch in the {counter} pass"
            got[counter] = (total / len(launches), len(launches))
    traffic = int(got["FETCH_SIZE"][0] * 1024 * 2 + got["WRITE_SIZE"][0] * 1024)
    return traffic, (f"live: this command re-executed under rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, "
                     f"{got['FETCH_SIZE'][1]} launches of {kern}* each, {a.batch} frames per launch); FETCH_SIZE KB x 1024 x 2 (gfx950 "
                     f"tallies 128-byte requests at 64) + WRITE_SIZE KB x 1024: read {int(got['FETCH_SIZE'][0] * 2048)} B + written "
                     f"{int(got['WRITE_SIZE'][0] * 1024)} B per launch")


def compulsory_bytes(H, W, n, B, dense, detector="m3"):
    """Bytes ONE launch of the dominant pixel kernel must move for a pass of B frames: every frame byte once plus the window
    history per time chunk (the launch geometry of launch_wide / run_pixels in metdet_hip.hip is restated here), plus what the
    dense output formats write.  Returns (bytes, time chunks, bit words per frame).  tools/profiles_table.py prices the committed
    rocprofv3 summaries with the same function."""
    frame = H * W
    words = H * ((W + 63) // 64)
    if detector == "classic":  # launch geometry of run_pixels (metdet_hip.hip)
        waves = -(-words // 16)
        nchunks = max(1, min(-(-8192 // waves), B // 8))
    else:                      # launch_wide: 16 pixels per lane up to n = 16
        waves = -(-words // 16) if W % 64 == 0 and n <= 16 else -(-words // 8)  # (n = 17 .. 32: the 8-pixels-per-lane variants)
        target = 16384 if B * frame <= 3.0e9 else 4096  # (launch_wide: short passes are cut into four times as many time chunks)
        nchunks = max(1, min(-(-target // waves), B // (2 * n)))
    tc = -(-B // nchunks)
    if detector != "classic" and W % 64 == 0 and n <= 32:
        tc = -(-tc // n) * n  # (round 6: every time chunk but the last is a whole number of window periods, launch_wide)
    if detector == "classic" and not dense and n == 4:
        tc = -(-tc // 4) * 4  # (k_classic_diff4_sparse: a time chunk is a whole number of four-frame bursts)
    nchunks = -(-B // tc)
    compulsory = frame * (B + nchunks * (n - 1))
    if dense and detector != "classic":
        compulsory += words * 8 * B  # the dense output format writes every word of the thresholded difference (1 bit per pixel)
    if detector == "classic" and dense:
        compulsory += 2 * words * 8 * B  # dense format: k_classic_diff_tb stores TWO plain bit planes per frame (|f1 - f0| > T and |f3 - f2| > T), every word
    return compulsory, nchunks, words


def roofline(prof_pipe, prof_alone, H, W, n, B, ms_per_pass, workload, dense=False, live_traffic=None):
    """Physical roofline of the dominant pixel kernel (the window kernel; Classic: the difference kernel).  Bytes one launch
    MUST move: every frame of the pass once plus the n-1 frames of history its window starts from, per time chunk (the
    kernel is blocked over the frames: grid.y chunks each re-read n-1 frames).  What it stores (tagged words of the sparse
    difference image: measured, ~0.6 % of the reads) is not counted.

    `frac` is priced on the NON-OVERLAPPED launch: the kernel's duration with one pass in flight, measured live in this run with
    HIP events on the library's stream (the `--standalone-passes` leg; `rocprofv3 --kernel-trace --stats` of `bench.py --depth 1`
    averages the same launches: profiles/*_kernel_stats_*_depth1.csv).  Inside the pipelined timed region two launches of the
    kernel overlap on the two main streams and share the machine with the other stages of the neighbouring passes, so a
    launch's span there is about twice as long without the kernel being any slower: that figure is kept as
    `frac_in_pipeline_per_launch` (what rocprofv3 averages for the default command), and `frac_at_pipeline_pace` prices the same
    bytes on the time the pipeline spends per pass -- the kernel's real share of the peak in the product."""
    kern = "classic" if DETECTOR == "classic" else "stack_threshold"
    out = {"bound": "hbm", "kernel": kern, "peak": HBM_PEAK_GBS, "unit": "GB/s"}
    ms, launches = prof_pipe.get(kern, (0.0, 0))
    if not launches:
        return None
    compulsory, nchunks, words = compulsory_bytes(H, W, n, B, dense, DETECTOR)
    frame = H * W
    avg_ms = ms / launches
    frac_of = lambda t_ms: compulsory / (t_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    out.update({
        "bytes_per_launch": compulsory, "frames_per_launch": B,
        "bytes_model": f"H*W*(frames + chunks*(n-1)) = {frame}*({B}+{nchunks}*{n - 1}): each frame byte once per launch plus the "
                       "window history per time chunk"
                       + (f"; + 2*{words}*8*{B} bytes written: the difference kernel stores two plain bit planes per frame (dense format)" if DETECTOR == "classic" and dense
                          else "; sparse format: only the words with set bits are stored (tagged), not counted" if DETECTOR == "classic"
                          else f"; + {words}*8*{B} bytes written: the run was in the DENSE output format (every word of the bit image stored)"
                          if dense else "; stores (sparse tagged words, ~0.6 % of the reads: measured) not counted"),
        "launch_ms_in_pipeline": avg_ms, "launches_in_pipeline": launches,
        "frac_in_pipeline_per_launch": frac_of(avg_ms),
        "launches_overlapping_in_pipeline": avg_ms / ms_per_pass,
        "in_pipeline_note": "mean span of the kernel's launches in the timed region (HIP events on the library's main streams, every "
                            "4th pass): consecutive launches overlap on the two main streams (MDH_MAIN_STREAMS=1 serialises them: "
                            "shorter spans, ~10 % fewer frames/s) next to the other stages of the neighbouring passes",
        # THE figures of the line: the kernel's bytes over the time the timed region spent per pass -- tied to
        # ms_per_step, the one number the driver's clock checks.  The non-overlapped launch is kept as `frac_isolated`.
        "achieved": compulsory / (ms_per_pass * 1e-3) / 1e9, "frac": frac_of(ms_per_pass),
        "frac_at_pipeline_pace": frac_of(ms_per_pass),
        "frac_note": "frac = bytes_per_launch / ms_per_pass / peak: priced on the pipelined timed region itself (ms_per_pass = ms_per_step / "
                     "passes_per_step); frac_isolated = the same bytes over the launch's own duration with one pass in flight (launch_ms)",
        "algorithmic_bytes_per_launch": (n + 1) * frame * B,
        "algorithmic_note": "SURVEY 8d's (n+1) reads per pixel per frame; the kernel keeps the window in registers and reads "
                            "each byte once per launch, so the figure priced on these bytes exceeds 1 by design and is not a bandwidth",
    })
    if prof_alone and prof_alone.get(kern, (0, 0))[1]:
        a_ms = prof_alone[kern][0] / prof_alone[kern][1]
        out.update({"launch_ms": a_ms, "launches": prof_alone[kern][1], "achieved_isolated": compulsory / (a_ms * 1e-3) / 1e9, "frac_isolated": frac_of(a_ms),
                    "launch_ms_note": "mean duration of the kernel's launches with ONE pass in flight (nothing overlaps), HIP events on "
                                      "the library's main stream, measured after the timed region of this run",
                    "algorithmic_frac": (n + 1) * frame * B / (a_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    else:  # --standalone-passes 0 (or --depth 1, where the timed region itself is non-overlapped)
        out.update({"launch_ms": avg_ms, "launches": launches, "achieved_isolated": compulsory / (avg_ms * 1e-3) / 1e9, "frac_isolated": frac_of(avg_ms),
                    "launch_ms_note": "mean span of the kernel's launches in the timed region (no stand-alone leg in this run)",
                    "algorithmic_frac": (n + 1) * frame * B / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
    traffic, src = None, None
    if live_traffic is not None and live_traffic[0] is not None:
        traffic, src = live_traffic
    else:
        tpath = os.path.join(ROOT, "profiles", f"pmc_traffic_{workload}.json")
        if os.path.exists(tpath):  # rocprofv3 PMC passes of this workload (tools/profile_round.sh), scaled to this batch
            t = json.load(open(tpath))
            if t.get("stack_threshold_bytes_per_launch"):
                traffic = int(t["stack_threshold_bytes_per_launch"] * B / t.get("frames_per_launch", B))
                src = (f"profiles/pmc_traffic_{workload}.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE in separate "
                       f"passes of this command at {t.get('frames_per_launch', B)} frames per launch (tools/profile_round.sh), scaled to "
                       "this batch -- NOT measured in this run"
                       + (f" ({live_traffic[1]})" if live_traffic is not None else ""))
    if traffic is not None:
        out["traffic_over_compulsory"] = traffic / compulsory
    out["traffic"] = traffic
    out["traffic_source"] = src
    return out


def run_streams(a, D, wl, mask):
    from metdetpy_amd import M3Detector
    from metdetpy_amd import _lib
    H, W, n = wl["H"], wl["W"], wl["n"]
    B = a.batch
    frames = make_stream_frames(wl, D.rank, mask, B, a.unique)
    uniq = len(frames)
    det = make_detector(n, mask, D.local, B)
    ctx = det._ctx
    if a.host_input:
        import ctypes
        pool = ctx.host_alloc(B * H * W)
        for start in range(0, B, uniq):
            cnt = min(uniq, B - start)
            ctypes.memmove(pool + start * H * W, frames[:cnt].ctypes.data, cnt * H * W)
    else:
        pool = ctx.device_alloc(B * H * W)
        for start in range(0, B, uniq):  # resident batch = the stream's period repeated
            cnt = min(uniq, B - start)
            ctx.device_upload(pool + start * H * W, frames[:cnt])
    info = np.zeros(B, _lib.FRAME_INFO_DTYPE)
    raw = np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32)
    nms = np.zeros((B, _lib.MDH_MAX_LINES, 4), np.int32)
    prob = np.zeros((B, _lib.MDH_MAX_LINES), np.float64)
    tally = [0, 0, 0]
    passes_seen = {}  # distinct per-pass tallies -> how many passes gave them (the timed region: ONE entry in steady state)
    # --with-collector: the pass's results go on to the native collector on a second host thread (the detector's host thread keeps
    # submitting and collecting); a ring of result buffers, so that the collector reads pass p while pass p + 1 is collected
    col = col_thread = col_q = None
    col_state = {"passes": 0, "busy_s": 0.0}
    ring = [(info, raw, nms, prob)]
    if a.with_collector:
        import queue
        import threading
        from metdetpy_amd.collector import MeteorCollector
        from metdetpy_amd.metstruct import CollectorCfg, MeteorCfg, RecheckSwitch, RuntimeParams
        ccfg = CollectorCfg(MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})  # config/m3det_normal.json's collector block
        rt = RuntimeParams(fps=25.0, exp_frame=1, eq_fps=25.0, eq_int_fps=25, exp_time=0.04, runtime_size=[W, H], raw_size=[W, H],
                           positive_category_list=["METEOR", "RED_SPRITE"])
        col = MeteorCollector(ccfg, rt)
        ring += [(np.zeros_like(info), None, np.zeros_like(nms), np.zeros_like(prob)) for _ in range(3)]
        col_q = queue.Queue(maxsize=len(ring) - 2)

        def col_worker():
            while True:
                item = col_q.get()
                if item is None:
                    return
                base, (i_, _r, n_, p_) = item
                t_ = time.perf_counter()
                col.update_batch(base, base, i_, n_, p_)
                col_state["busy_s"] += time.perf_counter() - t_
                col_state["passes"] += 1
                col_q.task_done()
        col_thread = threading.Thread(target=col_worker, name="bench-collector", daemon=True)
        col_thread.start()
    seen = {"passes": 0}

    def collect():
        bufs = ring[seen["passes"] % len(ring)]
        ctx.collect_into(bufs[0], bufs[1], bufs[2], bufs[3])
        info_ = bufs[0]
        one = (int(info_["n_raw"].sum()), int(info_["n_nms"].sum()), int(info_["lit_count"].sum()))
        passes_seen[one] = passes_seen.get(one, 0) + 1
        for k_ in range(3):
            tally[k_] += one[k_]
        if col is not None:
            col_q.put((seen["passes"] * B, bufs))  # (blocks when the collector is two passes behind: its pace then is the pipeline's)
        seen["passes"] += 1

    P = a.passes_per_step

    def run(npasses, depth=None):
        """npasses passes, up to ctx.max_in_flight queued: later passes are submitted before earlier ones are collected,
        so the pixel stages of one pass overlap the Hough stage of the previous one and the host stays ahead of the
        GPU.  Every pass is fully collected before run() returns."""
        depth = depth or (a.depth if a.depth > 0 else ctx.max_in_flight)
        for _s in range(npasses):
            if ctx.in_flight >= depth:
                collect()
            ctx.submit(pool, B, not a.host_input)
        while ctx.in_flight:
            collect()
        if col is not None:
            col_q.join()  # a run ends when the collector has seen its last pass

    if a.warmup:
        run(a.warmup * P)
    col_state.update(passes=0, busy_s=0.0)
    steps = a.steps if a.steps is not None else 20
    tally[:] = [0, 0, 0]
    passes_seen.clear()
    D.fence()
    t0 = time.perf_counter()
    run(steps * P)  # the timed region carries no timing events: they cost the pipeline a few per cent
    D.fence()
    segs, kept, lit = tally
    dt = time.perf_counter() - t0
    timed = {"tally": [segs, kept, lit], "frames": steps * P * B, "per_pass_equal": len(passes_seen) == 1, "warm_frames": a.warmup * P * B, "pool_frames": B}
    prof = {}
    if not a.no_profile and D.rank == 0:
        # the kernels' spans INSIDE the pipeline (overlapping each other, queueing included): the same pipelined passes again,
        # untimed, every 4th one carrying HIP events on the library's streams
        ctx.profile_enable(4)
        ctx.profile_read(reset=True)
        run(max(8, min(steps * P, 24)))
        prof = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
    prof_alone = {}
    if not a.no_profile and a.standalone_passes > 0 and D.rank == 0:
        run(2, depth=1)
        ctx.profile_enable(1)
        ctx.profile_read(reset=True)
        run(a.standalone_passes, depth=1)  # one pass in flight: nothing overlaps, the spans are the kernels' own durations
        prof_alone = ctx.profile_read(reset=True)
        ctx.profile_enable(False)
    dt_all, tot = D.max_time_and_sum(dt, [segs, kept, lit])
    res = dict(frames=steps * P * B * D.world, dt=dt_all, steps=steps, tot=tot, prof=prof, prof_alone=prof_alone, frames_sample=frames,
               B=B, P=P, uniq=uniq, dense=bool(ctx.dense_now), timed=timed)
    if col is not None:
        timed_passes, busy = steps * P, col_state["busy_s"]
        col_q.put(None)
        col_thread.join(30)
        recs = col.met_exporter.meteor_list
        res["extra"] = {"collector": {
            "in_timed_region": True, "thread": "second host thread, mdh_collector_update_batch per collected pass (ring of 4 result buffers)",
            "records_exported_during_the_run": len(recs), "series_active_at_the_end": int(col.num_active),
            "records_note": "the bench stream carries a streak every 50 frames for millions of frames: at 25 frames per second that is inside "
                            "max_interval (4 s), so there is always an active series and the waiting list is never flushed (collector.py:463-480) "
                            "-- the series pile up as ONE open group, which is the reference's behaviour on such a stream; the records themselves "
                            "are compared with the oracle's on the parity frames (parity.records_identical)",
            "collector_busy_s_all_passes": busy, "collector_passes": col_state["passes"],
            "collector_alone_frames_per_s": (col_state["passes"] * B / busy) if busy else None,
            "note": "value = frames per second with every pass's results consumed by the native collector before the region ends; "
                    "collector_alone = what the collector thread's own busy time would allow; records identical to the oracle's: parity.records"}}
        del timed_passes
    (ctx.host_free if a.host_input else ctx.device_free)(pool)
    det.close()
    return res


def run_per_frame(a, D, wl, mask):
    """The reference's own loop (MetDetPy.py:185-202): detector.update(frame); detector.detect() per frame, frames in
    host memory, results on the host before the next frame is handed over.  One step = one frame.

    --frames says where the loader's frames live:
      pinned    a pool of mdh_host_alloc (a loader built for this detector): every frame is read in place by the window kernel
      recycled  a ring of 8 ordinary NumPy buffers refilled in turn -- ProcessVideoLoader's shared RawArray ring
                (videoloader.py:696-701, 746-772).  The refill (one host memcpy per frame, the loader's decode stand-in) happens
                OUTSIDE the timed calls: only update() + detect() are timed, as for `pinned`.  The loader opts in to registration
                (mdh_set_auto_register: off by default): the library registers each buffer the third time it comes by and reads it
                in place from then on; no call to host_alloc
      pageable  a fresh NumPy array per frame (ThreadVideoLoader, videoloader.py:569-618): the library's default, nothing is
                registered, every frame takes the copy path (H2D from pageable memory)
    The per-frame time is also reported split into frames without / with raw Hough segments (a streak costs a serial replay)."""
    from metdetpy_amd import M3Detector
    H, W, n = wl["H"], wl["W"], wl["n"]
    import ctypes
    frames = make_stream_frames(wl, D.rank, mask, 300, a.unique)
    det = M3Detector(1, n, mask, 10, make_cfg(), None, device=D.local, max_batch=8)
    host_frames = frames
    pool = None
    if a.frames == "pinned":
        # the loader's frame buffers: pinned host memory (mdh_host_alloc), as a loader built for this detector would hand them over
        pool = det._ctx.host_alloc(frames.nbytes)
        ctypes.memmove(pool, frames.ctypes.data, frames.nbytes)
        frames = np.ctypeslib.as_array((ctypes.c_uint8 * frames.nbytes).from_address(pool)).reshape(frames.shape)
        source = lambda k: frames[k % len(frames)]
    elif a.frames == "recycled":
        ring = [np.empty((H, W), np.uint8) for _ in range(8)]
        det._ctx.set_auto_register(True)  # the loader's opt-in: it owns a fixed ring that outlives the detector

        def source(k):
            b = ring[k % len(ring)]
            np.copyto(b, host_frames[k % len(host_frames)])  # the loader refills the buffer (untimed: stands for its decode)
            return b
    else:
        # (the whole period, like the other two legs: the same seamless stream, the same share of streak frames -- up to round 5's
        # first collection this leg cycled the first 64 frames, 44 % of them with a streak against the stream's 36 %)
        fresh = [host_frames[i].copy() for i in range(len(host_frames))]
        source = lambda k: fresh[k % len(fresh)]
    tally = [0, 0, 0]
    split = {"quiet": [0, 0.0], "streak": [0, 0.0]}
    k = 0

    def run(nsteps, timed=False):
        nonlocal k
        spent = 0.0
        for _ in range(nsteps):
            f = source(k)
            t0 = time.perf_counter()
            det.update(f)
            lines, _pred = det.detect()
            dt = time.perf_counter() - t0
            spent += dt
            tally[0] += det.lines_num
            tally[1] += len(lines)
            if timed:
                e = split["streak" if det.lines_num else "quiet"]
                e[0] += 1
                e[1] += dt
            k += 1
        return spent

    run(max(a.warmup, 3 * n + 24))
    steps = a.steps if a.steps is not None else 20000
    tally[:] = [0, 0, 0]
    D.fence()
    t0 = time.perf_counter()
    spent = run(steps, timed=True)
    D.fence()
    dt = time.perf_counter() - t0
    if a.frames == "recycled":
        dt = spent  # update() + detect() only: the loader's refill of the ring buffer is not the detector's time
    dt_all, tot = D.max_time_and_sum(dt, tally)
    stats = det._ctx.update_stats()
    extra = {"per_frame": {"frames": a.frames, "update_stats": stats,
                           "us_per_frame": {k_: (v[1] / v[0] * 1e6 if v[0] else None) for k_, v in split.items()},
                           "frames_by_kind": {k_: v[0] for k_, v in split.items()},
                           "frames_per_s_by_kind": {k_: (v[0] / v[1] if v[1] else None) for k_, v in split.items()},
                           "note": "quiet: frames without raw Hough segments; streak: frames with (their replay is a serial chain). The blended "
                                   "`value` depends on the stream's share of streak frames"}}
    del frames
    if pool is not None:
        det._ctx.host_free(pool)
    det.close()
    return dict(frames=steps * D.world, dt=dt_all, steps=steps, tot=tot, prof={}, prof_alone={}, frames_sample=host_frames, B=1,
                uniq=len(host_frames), extra=extra)


SECONDARY = [
    # (name, extra arguments): each >= 1 s timed at the rates of profiles/, parity on 64 frames, no CPU baseline, no PMC leg
    ("C3_4k_n16_mask", ["--workload", "C3", "--unique", "100", "--steps", "13", "--warmup", "2"]),
    # (1000 frames per pass, 12 passes per step -- as many frames per step as the workload's default 500 x 24: 127 k -> 134 k frames/s on one box)
    ("C5_8k_high", ["--workload", "C5", "--unique", "50", "--batch", "1000", "--passes-per-step", "12", "--steps", "13", "--warmup", "2"]),
    ("C1s_960x540", ["--workload", "C1s", "--steps", "17", "--warmup", "2"]),
    # the reference's own default window (config/m3det_normal.json: window_sec 1 -> n = int(window_sec * fps) = 25 / 30, Detector.py:200)
    ("C2_n25_reference_default_window", ["--window", "25", "--unique", "100", "--steps", "8", "--warmup", "1"]),
    # ... and the same window on the stream with a streak in 5 % of the frames instead of 26 %: at n = 25 a streak stays in the window
    # for 37 frames, so the default stream has points for the Hough stage in 70 % of its frames -- this leg is the window kernel's pace
    # when the Hough load is a meteor camera's, not a stress test's
    ("C2_n25_streak_share_5pct", ["--window", "25", "--stream", "sparse", "--unique", "400", "--steps", "10", "--warmup", "1"]),
    ("C1s_960x540_n25", ["--workload", "C1s", "--window", "25", "--unique", "100", "--steps", "8", "--warmup", "1"]),
    ("C1s_960x540_n30", ["--workload", "C1s", "--window", "30", "--unique", "100", "--steps", "8", "--warmup", "1"]),
    # what a consumer with a latency budget sees: the headline workload in passes of 1000 and 250 frames
    ("C2_pass_1000_frames", ["--batch", "1000", "--passes-per-step", "128", "--steps", "8", "--warmup", "1"]),
    ("C2_pass_250_frames", ["--batch", "250", "--passes-per-step", "256", "--steps", "8", "--warmup", "1"]),
    # the headline workload with the native collector inside the timed region, on a second host thread (item 7)
    ("C2_with_collector", ["--with-collector", "--steps", "8", "--warmup", "1"]),
    # the unfriendly skies: 1080p, a 100-frame period (sparse: 400, dense: 96), parity + worst-frame replay time each
    ("C2_noisy_sky_m3det_low", ["--stream", "noisy", "--unique", "100", "--steps", "6", "--warmup", "1"]),
    ("C2_hot_pixels_1pct", ["--stream", "hot", "--unique", "100", "--steps", "6", "--warmup", "1"]),
    ("C2_cloud_edge_every_50", ["--stream", "cloud", "--unique", "100", "--steps", "3", "--warmup", "1", "--passes-per-step", "16"]),
    ("C2_streak_share_5pct", ["--stream", "sparse", "--unique", "400", "--steps", "6", "--warmup", "1"]),
    ("C2_streak_share_80pct", ["--stream", "dense", "--unique", "96", "--batch", "3840", "--steps", "6", "--warmup", "1"]),
    # BASELINE config 5 the way it is sharded (8 row bands), all bands on this one device: prices the tiling's bookkeeping
    ("C5_8k_as_8_row_bands_one_gpu", ["--mode", "bands", "--workload", "C5", "--unique", "50", "--batch", "500", "--steps", "12", "--warmup", "3"]),
    ("per_frame_pinned", ["--protocol", "per-frame", "--frames", "pinned", "--steps", "12000"]),
    ("per_frame_recycled_ring", ["--protocol", "per-frame", "--frames", "recycled", "--steps", "12000"]),
    ("per_frame_pageable", ["--protocol", "per-frame", "--frames", "pageable", "--steps", "4000"]),
]


def run_secondary(a):
    """The other workloads and the per-frame protocol, by this same script in subprocesses (fresh HIP state each; the headline
    detector has been closed).  Each line is parsed and reduced to the figures VERDICT asks for; the full lines go to stderr."""
    import subprocess
    out = {}
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for name, extra in SECONDARY:
        cmd = [sys.executable, os.path.abspath(__file__), "--no-secondary", "--no-pmc", "--cpu-seconds", "0", "--parity-frames", "64",
               "--standalone-passes", "6"] + extra
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=480)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if p.returncode != 0 or not line:
                out[name] = {"error": f"rc {p.returncode}: {p.stderr.strip()[-300:]}"}
                continue
            d = json.loads(line[-1])
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            continue
        print(f"[secondary {name}] {line[-1]}", file=sys.stderr)
        rf = d.get("roofline") or {}
        e = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "window": d["config"].get("window"),
             "frac_isolated": rf.get("frac_isolated"),
             "timed_s": d["ms_per_step"] * d["steps"] * 1e-3,
             "frames_per_step": d["config"]["frames_per_step"], "passes_per_step": d["config"].get("passes_per_step"),
             "frac": rf.get("frac"), "frac_at_pipeline_pace": rf.get("frac_at_pipeline_pace"),
             "parity": (d.get("parity") or {}).get("verdict"), "parity_frames": (d.get("parity") or {}).get("frames_checked"),
             "timed_region_tallies": ((d.get("parity") or {}).get("timed_region") or {}).get("verdict"),
             "hough_frame_cost": (d.get("parity") or {}).get("hough_frame_cost"),
             "wall_s": round(time.perf_counter() - t0, 1), "command": "python bench.py " + " ".join(cmd[2:])}
        if "per_frame" in d:
            e["per_frame"] = {k: d["per_frame"][k] for k in ("frames", "update_stats", "us_per_frame", "frames_by_kind", "frames_per_s_by_kind")}
        if "collector" in d:
            e["collector"] = {k: d["collector"].get(k) for k in ("records_exported_during_the_run", "series_active_at_the_end", "collector_alone_frames_per_s")}
            e["records_identical"] = (d.get("parity") or {}).get("records_identical")
        out[name] = e
        time.sleep(2.0)  # (a child that held ~100 GB of HBM has just exited: the driver releases -- and wipes -- its memory in the background)
    return out


def orchestrate(a):
    """The plain headline command on one GPU: this process never touches HIP.  The headline region runs in a child (this same
    script with --no-secondary: barrier, timed steps, roofline, cpu_baseline, parity -- its JSON line is taken as it is), then the
    `secondary` legs run in children of their own, and ONE line is printed.  Round 4 ran the headline here and the legs as
    children of a parent that had released the device (hipDeviceReset) but stayed alive: the 8K legs ran 5-25 % below what the
    same commands reach on their own (C5 108 k against 113 k, the row bands 63 k against 84 k frames/s, same box)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + ["--no-secondary"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not lines:
        sys.stdout.write(p.stdout)
        sys.exit(p.returncode or 1)
    head = json.loads(lines[-1])
    time.sleep(2.0)
    sec = run_secondary(a)
    # The driver keeps the LAST 16 KB of this process's output (round 5: its tail began in the middle of the line and the C3 / C5 /
    # C1s legs were cut off).  So the line is ordered for its tail: the bulky objects first, then the headline's own figures, and
    # last of all `secondary_summary` -- leg -> [frames/s, frac at the pipeline's pace, isolated frac, parity | timed-region tallies] in
    # well under 2 KB.  The same summary is also printed as a line of its own in front of the JSON line.
    def short(v):
        return None if v is None else ("ok" if v == "identical" else str(v)[:24])
    summary = {k: ([round(v["value"]), None if v.get("frac") is None else round(v["frac"], 3),
                    None if v.get("frac_isolated") is None else round(v["frac_isolated"], 3), short(v.get("parity")), short(v.get("timed_region_tallies"))]
                   if "error" not in v else ["error", v["error"][:60]]) for k, v in sec.items()}
    bulky = ("cpu_baseline", "parity", "kernels_ms_per_pass", "kernels_ms_standalone", "detections")
    out = {k: head[k] for k in bulky if k in head}
    out["secondary"] = sec
    out.update({k: v for k, v in head.items() if k not in bulky})
    out["secondary_summary_columns"] = ["frames_per_s", "frac_at_pipeline_pace", "frac_isolated", "parity", "timed_region_tallies"]
    out["secondary_summary"] = summary
    print("[secondary_summary] " + json.dumps(summary), flush=True)
    print(json.dumps(out), flush=True)


def main():
    sys.modules.setdefault("bench", sys.modules[__name__])  # (bench_modes does `import bench`: the module this script runs as, not a second copy)
    a = parse()
    if (int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("MDH_BENCH_FORCE_DIST") != "1" and not a.no_secondary and a.gpus == 1
            and a.mode == "streams" and a.protocol == "batch" and a.workload == "C2" and a.stream == "sky" and a.detector == "m3"
            and a.hough_mode == "exact" and not a.host_input and a.depth == 0 and a.sensitivity is None and a.window is None):
        return orchestrate(a)
    D = Dist()
    wl = dict(WORKLOADS[a.workload])
    if a.window is not None:
        wl["n"] = a.window
        old_text = f"{WORKLOADS[a.workload]['n']}-frame window"
        wl["desc"] = (wl["desc"].replace(old_text, f"{a.window}-frame window (--window)") if old_text in wl["desc"]
                      else wl["desc"] + f", {a.window}-frame window (--window)")
    H, W, n = wl["H"], wl["W"], wl["n"]
    global SENSITIVITY, DETECTOR, HOUGH_MODE, STREAM
    STREAM = a.stream
    SENSITIVITY = a.sensitivity or STREAMS[STREAM].get("sensitivity") or wl.get("sensitivity", "normal")
    DETECTOR = a.detector
    HOUGH_MODE = a.hough_mode
    if DETECTOR == "classic":
        n = 4
    if a.batch is None:
        a.batch = wl["batch"]
    plain_streams = a.mode == "streams" and a.protocol == "batch" and a.workload != "C1e"
    if a.passes_per_step is None:
        a.passes_per_step = wl.get("pps", 1) if plain_streams else 1
    if not plain_streams:
        a.passes_per_step = 1
    if a.protocol == "per-frame" and a.warmup == 2:
        a.warmup = 6
    # (--mode bands with all bands in one process: LocalBands asks the binding for 4 hardware queues itself, _lib.prefer_hw_queues)
    mask = np.ones((H, W), np.uint8)
    if a.workload in ("C3", "C1e"):
        m = load_mask(H, W)
        if m is not None:
            mask = m
    extra = {}
    if a.workload == "C1e":
        import bench_modes
        r = bench_modes.run_c1e(a, D, wl, mask, make_cfg())
        extra = r.pop("extra", {})
    elif a.mode != "streams":
        import bench_modes
        r = (bench_modes.run_bands if a.mode == "bands" else bench_modes.run_timeshard)(a, D, wl, mask, make_cfg())
        extra = r.pop("extra", {})
    elif a.protocol == "per-frame":
        r = run_per_frame(a, D, wl, mask)
        extra = r.pop("extra", {})
    else:
        r = run_streams(a, D, wl, mask)
        extra = r.pop("extra", {})
    fps = r["frames"] / r["dt"]
    B = r["B"]
    P = r.get("P", 1)
    shape = {"streams": f"one independent stream per GPU, {P} passes of {B} frames per step, frames resident in HBM",
             "bands": f"ONE stream tiled into row bands, {B} frames/step", "timeshard": f"ONE stream cut into time windows, {B} frames/step"}[a.mode]
    if a.protocol == "per-frame":
        shape = f"update(frame) / detect() per frame from host memory (the reference's loop; frames: {a.frames}), one frame per step"
    desc = wl["desc"]
    if STREAM != "sky":
        desc += f", --stream {STREAM} ({json.dumps(STREAMS[STREAM]['kw'])})"
    if a.mode == "bands":  # (the workload's own text says "whole frame on one GPU (no row-band tiling)": not this run)
        desc = desc.replace(", whole frame on one GPU (no row-band tiling)", "")
    out = {
        "metric": "frames/sec at 1920x1080 (1/2/4/8 GPU) + detection parity vs CPU ref",
        "value": fps,
        "unit": "frames/s",
        "n_gpus": D.world,
        "steps": r["steps"],
        "warmup": a.warmup,
        "ms_per_step": r["dt"] / r["steps"] * 1e3,
        "higher_is_better": True,
        "scaling": "weak" if a.mode == "streams" else "strong",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": f"{a.workload}: {desc}, "
                               + (f"ClassicDetector (4-frame ring, fixed threshold {CLASSIC_THRESHOLD})" if DETECTOR == "classic" else f"m3det_{SENSITIVITY} detector cfg")
                               + (", hough_mode = accumulator (NOT cv2.HoughLinesP's segments)" if HOUGH_MODE == "accumulator" else "")
                               + f", {shape}"
                               + (" -- INPUT FROM PINNED HOST MEMORY (PCIe-inclusive)" if a.host_input else ""),
                   "frame": [W, H], "window": n, "frames_per_step": B * P, "passes_per_step": P, "frames_per_pass": B,
                   "stream_period_frames": r["uniq"],
                   "parallelism": f"{a.mode} x{D.world}" if a.mode != "streams" else f"streams x{D.world}"},
        "detections": {"raw_segments": r["tot"][0], "nms_segments": r["tot"][1], "lit_pixels": r["tot"][2]},
    }
    if D.g is not None:
        out["config"]["collectives"] = (f"{D.backend}: " + ("ncclAllReduce(max) + ncclAllGather on a private HIP stream (librccl through ctypes), "
                                        "ncclUniqueId and barrier through the rendezvous store" if D.backend == "rccl" else
                                        "ranks share devices (RCCL refuses two ranks per device): barrier, max and gather through the rendezvous store"))
        if a.mode == "streams" and getattr(D, "per_rank", None):
            out["detections"]["per_rank_raw_segments"] = [row[0] for row in D.per_rank]
    out.update(extra)
    if (D.world > 1 and a.mode == "streams" and a.protocol == "batch" and a.parity_frames > 0 and r.get("frames_sample") is not None
            and DETECTOR == "m3" and HOUGH_MODE == "exact" and a.workload != "C1e"):
        # N > 1: every rank checks ITS OWN stream (seed + rank) against the oracle -- first periods frame by frame and the timed
        # region's tallies, like the one-GPU line -- on its own host cores, at the same time; rank 0 prints its own object in full
        # and the verdicts of all ranks.  (A rank that fails says so instead of leaving the others waiting.)
        try:
            mine = parity_check(r["frames_sample"], n, mask, D.local, a.parity_frames, timed=r.get("timed") if a.warmup > 0 else None)
        except Exception as e:  # noqa: BLE001
            mine = {"verdict": f"error: {type(e).__name__}: {e}"}
        every = D.all_gather_object({"rank": D.rank, "verdict": mine.get("verdict"),
                                     "timed_region": (mine.get("timed_region") or {}).get("verdict")})
        if D.rank == 0:
            mine["per_rank"] = every
            verdicts = {e["verdict"] for e in every}
            if len(verdicts) > 1 or mine.get("verdict") not in verdicts:
                mine["verdict"] = "MISMATCH on some rank: " + json.dumps(sorted(str(v) for v in verdicts))
            out["parity"] = mine
    if D.rank == 0:
        if r["prof"]:
            live = None
            if (not a.no_pmc and D.world == 1 and a.mode == "streams" and a.protocol == "batch" and a.workload != "C1e"
                    and not a.host_input):
                live = pmc_traffic_live(a, bool(r.get("dense")))
            out["ms_per_pass"] = out["ms_per_step"] / P
            rf = roofline(r["prof"], r["prof_alone"], H, W, n, B, out["ms_per_pass"], a.workload, dense=bool(r.get("dense")),
                          live_traffic=live)
            if rf:
                out["roofline"] = rf
            out["kernels_ms_per_pass"] = {k: v[0] / v[1] for k, v in r["prof"].items() if v[1]}  # mean span per pass, in the pipeline
            if r["prof_alone"]:
                out["kernels_ms_standalone"] = {k: v[0] / v[1] for k, v in r["prof_alone"].items() if v[1]}
        frames = r["frames_sample"]
        if D.world == 1 and a.cpu_seconds > 0 and frames is not None and DETECTOR == "m3":
            out["cpu_baseline"] = cpu_baseline(frames, n, mask, a.cpu_seconds)
        if D.world == 1 and a.parity_frames > 0 and frames is not None and "parity" not in out:
            if a.protocol == "per-frame" and DETECTOR == "m3" and HOUGH_MODE == "exact":
                out["parity"] = parity_check_per_frame(frames, n, mask, D.local, a.parity_frames, a.frames)
            elif DETECTOR == "classic":
                out["parity"] = parity_check_classic(frames, mask, D.local, a.parity_frames)
            elif HOUGH_MODE == "accumulator":
                out["parity"] = parity_check_accumulator(frames, n, mask, D.local, a.parity_frames)
            else:
                out["parity"] = parity_check(frames, n, mask, D.local, a.parity_frames, timed=r.get("timed") if a.warmup > 0 else None)
        print(json.dumps(out), flush=True)
    D.close()


if __name__ == "__main__":
    main()
