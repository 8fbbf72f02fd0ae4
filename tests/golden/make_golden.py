"""Generate golden vectors from the REFERENCE ITSELF (run in the build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports the reference's own NumPy-only classes from /root/reference (read-only) with stub modules
for the third-party packages that are absent from this image (cv2, onnxruntime, dacite, av, rawpy,
pyexiv2, ...).  No reference source is copied: only the inputs and the outputs the reference produced
are written, as small .npz/.json fixtures next to this script.  Frames are regenerated in the tests
from the recorded seeds, so they are not stored.

Fixtures
  g1_sliding_window.npz   SlidingWindow.update -> max / mean / sum / length per step    (utils.py:225-307)
  g2_threshold_series.json LineDetector.update threshold / snr series                   (Detector.py:73-91, 225-229)
  g3_std_roi.json         SNR_SW.select_subarea on synthetic masks                      (Detector.py:93-122)
  g4_lineset_nms.json     lineset_nms in/out (no length ties between different segments) (utils.py:780-839)
  g5_collector.json       MeteorCollector scripted (frame, lines, cates) -> MDRF records (collector.py:372-770)
  g6_time.json            frame2ts / ts2frame / time2frame / frame2time                  (utils.py:635-720)
  g7_ema.json             EMA.update series                                             (utils.py:324-368)
  g8_exposure.json        rf_estimator / _rf_est_kernel / sigma_clip / init_exp_time on scripted frame-sum series
                          (videoloader.py:440-494, 796-870; utils.py:516-540)
  g11_visu.json           M3Detector.visu(): the attribute records handed to the visual manager (Detector.py:394-448,
                          metvisu.py:74-260), as dataclasses.asdict, for scripted detector states
  g10_mdrf.json           the MDRF document detect_video returns (MetDetPy.py:245-250) as MDRF.to_dict() writes it
                          (metstruct.py:105-111, 428-441): version / basic_info / config / type / anno_size / results /
                          performance, for the reference's config/m3det_*.json presets and two scripted collector runs
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.__getattr__ = lambda name: 0  # constants such as cv2.MORPH_RECT
    cv2.getStructuringElement = lambda shape, ksize: np.ones(ksize[::-1], np.uint8)
    cv2_typing = types.ModuleType("cv2.typing")
    cv2_typing.MatLike = np.ndarray
    cv2.typing = cv2_typing
    ort = types.ModuleType("onnxruntime")
    ort.set_default_logger_severity = lambda *_: None
    ort.get_available_providers = lambda: ["CPUExecutionProvider"]
    dacite = types.ModuleType("dacite")
    dacite.from_dict = lambda data_class, data: (_ for _ in ()).throw(NotImplementedError)
    av = types.ModuleType("av")
    av_error = types.ModuleType("av.error")
    av.error = av_error
    for name, mod in [("cv2", cv2), ("cv2.typing", cv2_typing), ("onnxruntime", ort), ("dacite", dacite), ("av", av),
                      ("av.error", av_error), ("rawpy", types.ModuleType("rawpy")),
                      ("pyexiv2", types.ModuleType("pyexiv2")), ("multiprocess", None)]:
        if mod is None:
            continue
        sys.modules.setdefault(name, mod)
    sys.path.insert(0, REF)


def frames_normal(seed, count, shape, mu, sigma):
    rng = np.random.default_rng(seed)
    return [np.clip(np.rint(rng.normal(mu, sigma, shape)), 0, 255).astype(np.uint8) for _ in range(count)]


def main():
    install_stubs()
    from MetLib import utils as U
    from MetLib import Detector as D
    from MetLib import collector as CO
    from MetLib import metstruct as MS
    from MetLib.metlog import BaseMetLog

    # ---------------------------------------------------------------- G1
    g1 = {}
    shape = (24, 40)
    for n in (1, 4, 8):
        rng = np.random.default_rng(100 + n)
        frames = rng.integers(0, 256, (3 * n, ) + shape, dtype=np.uint8)
        sw = U.SlidingWindow(n, shape, dtype=np.uint8, force_int=True)
        mx, mean, sm, ln = [], [], [], []
        for f in frames:
            sw.update(f)
            mx.append(sw.max.copy())
            mean.append(np.array(sw.mean))
            sm.append(sw.sum.copy())
            ln.append(sw.length)
        g1[f"n{n}_seed"] = np.array(100 + n)
        g1[f"n{n}_max"] = np.array(mx)
        g1[f"n{n}_mean"] = np.array(mean)
        g1[f"n{n}_sum"] = np.array(sm)
        g1[f"n{n}_len"] = np.array(ln)
    np.savez_compressed(os.path.join(HERE, "g1_sliding_window.npz"), **g1)

    # ---------------------------------------------------------------- G2
    def binary_cfg(sens, adaptive=True, init_value=7, area=0.1, interval=2):
        return MS.BinaryCfg(binary=MS.BinaryCoreCfg(adaptive, init_value, sens, area, interval),
                            hough_line=MS.HoughLineCfg(10, 10, 10),
                            dynamic=MS.DynamicCfg(True, 5))

    g2 = []
    shape = (108, 192)
    mask = np.ones(shape, np.uint8)
    cases = [("normal", 8, 1.0, 2, 0.1), ("normal", 8, 2.5, 2, 0.1), ("high", 8, 2.5, 2, 0.1), ("low", 8, 6.0, 2, 0.1),
             ("normal", 5, 2.5, 1, 0.1), ("normal", 16, 4.0, 3, 0.25), ("high", 4, 0.4, 2, 0.5)]
    for ci, (sens, n, sigma, interval, area) in enumerate(cases):
        seed = 2000 + ci
        det = D.M3Detector(window_sec=1, fps=n, mask=mask, num_cls=10, cfg=binary_cfg(sens, area=area, interval=interval),
                           logger=BaseMetLog())
        steps = 12 * n * interval + 3
        rec = {"sensitivity": sens, "n": n, "sigma": sigma, "interval": interval, "area": area, "seed": seed,
               "shape": list(shape), "mu": 30, "steps": steps, "std_roi": [int(v) for v in det.stack.std_roi],
               "bi_threshold": [], "bi_threshold_float": [], "snr": []}
        for f in frames_normal(seed, steps, shape, 30, sigma):
            det.update(f)
            rec["bi_threshold"].append(int(det.bi_threshold))
            rec["bi_threshold_float"].append(float(det.bi_threshold_float))
            rec["snr"].append(float(det.stack.snr))
        g2.append(rec)
    # a non-adaptive case
    det = D.M3Detector(window_sec=1, fps=8, mask=mask, num_cls=10, cfg=binary_cfg("normal", adaptive=False, init_value=7),
                       logger=BaseMetLog())
    rec = {"sensitivity": "normal", "n": 8, "sigma": 2.5, "interval": 2, "area": 0.1, "seed": 2999, "adaptive": False,
           "init_value": 7, "shape": list(shape), "mu": 30, "steps": 20, "std_roi": [int(v) for v in det.stack.std_roi],
           "bi_threshold": [], "bi_threshold_float": [], "snr": []}
    for f in frames_normal(2999, 20, shape, 30, 2.5):
        det.update(f)
        rec["bi_threshold"].append(int(det.bi_threshold))
        rec["bi_threshold_float"].append(float(det.bi_threshold_float))
        rec["snr"].append(float(det.stack.snr))
    g2.append(rec)
    json.dump(g2, open(os.path.join(HERE, "g2_threshold_series.json"), "w"))

    # ---------------------------------------------------------------- G3
    g3 = []

    def roi_of(mask, area):
        sw = D.SNR_SW(n=2, mask=mask, est_snr=True, est_area=area, nz_interval=1)
        return [int(v) for v in sw.std_roi]

    for (h, w) in [(540, 960), (1080, 1920), (270, 480), (101, 203)]:
        for area in (0.1, 0.25):  # area == 0 crashes the reference (negative sub-window shape, Detector.py:106,65-68)
            for kind in ("ones", "bottom", "bottom_heavy", "corner", "stripe"):
                m = np.ones((h, w), np.uint8)
                if kind == "bottom":
                    m[int(h * 0.62):] = 0
                elif kind == "bottom_heavy":
                    m[int(h * 0.35):] = 0
                elif kind == "corner":
                    m[int(h * 0.5):, :int(w * 0.5)] = 0
                elif kind == "stripe":
                    m[:, int(w * 0.45):int(w * 0.5)] = 0
                g3.append({"h": h, "w": w, "area": area, "kind": kind, "roi": roi_of(m, area)})
    json.dump(g3, open(os.path.join(HERE, "g3_std_roi.json"), "w"))

    # ---------------------------------------------------------------- G4
    g4 = []
    rng = np.random.default_rng(4000)
    for case in range(60):
        m = int(rng.integers(1, 40))
        base = rng.integers(0, 400, (m, 2))
        delta = rng.integers(-60, 61, (m, 2))
        delta[(np.abs(delta).max(axis=1) < 10)] += 12  # PPHT output always spans >= min_len on one axis
        lines = np.concatenate([base, base + delta], axis=1).astype(np.int32)
        if case % 5 == 0 and m > 2:
            lines[1] = lines[0]  # exact duplicate
        if case % 7 == 0 and m > 3:
            lines[2, :2] = lines[3, :2] + 1
            lines[2, 2:] = lines[3, 2:] + 2  # near duplicate (different length)
        # Equal squared lengths between DIFFERENT segments make the reference's np.argsort(...)[::-1] order
        # implementation-defined (numpy's SIMD argsort is not stable), so such sets are not golden material.
        l2 = (lines[:, 2] - lines[:, 0])**2 + (lines[:, 3] - lines[:, 1])**2
        tie = any(l2[i] == l2[j] and not np.array_equal(lines[i], lines[j]) for i in range(m) for j in range(i))
        if tie:
            continue
        out, prob = U.lineset_nms(lines.copy())
        g4.append({"lines": lines.tolist(), "out": np.asarray(out).tolist(), "nonline": [float(p) for p in prob]})
    json.dump(g4, open(os.path.join(HERE, "g4_lineset_nms.json"), "w"))

    # ---------------------------------------------------------------- G5
    class CaptureLog(BaseMetLog):

        def __init__(self):
            self.rows = []

        def log(self, level, string):
            self.rows.append([int(level), string])

    U.get_num_class()

    def model_cfg():
        return MS.ModelCfg("YOLOModel", "./weights/yolov5s_v2.onnx", "float32", True, True, 0.25, 0.45, 2, 2)

    def collector_cfg(preset):
        if preset == "normal":
            met = MS.MeteorCfg(15, 4, [0, 8], [2, 21], [0, 0.6], 0.5, 2048)
        elif preset == "high":
            met = MS.MeteorCfg(15, 4, [0, 10], [1.5, 24], [0, 0.6], 0.5, 2048)
        else:
            met = MS.MeteorCfg(20, 5, [0, 8], [3, 18], [0, 0.6], 0.5, 2048)
        return MS.CollectorCfg(met, MS.RecheckCfg(False, model_cfg()), {"positive_cates": ["METEOR", "RED_SPRITE"]})

    def run_script(preset, rt, script, tail_every=None, total=None):
        log = CaptureLog()
        col = CO.MeteorCollector(collector_cfg(preset), rt, None, log)
        by_frame = {fr: (ln, ct) for fr, ln, ct in script}
        last = total if total is not None else (max(by_frame) + 1 if by_frame else 1)
        for i in range(0, last, rt.exp_frame):
            it = i // rt.exp_frame
            if i in by_frame:
                ln, ct = by_frame[i]
                col.update(i, lines=np.array(ln, dtype=np.int32).reshape(-1, 4), cates=np.array(ct, dtype=np.float64))
            elif it % rt.eq_int_fps == 0:
                col.update(i, lines=[], cates=[])
        col.clear()
        recs = [r.to_dict() for r in col.met_exporter.meteor_list]
        return recs, log.rows

    def m3_cates(nonline):
        c = np.zeros((len(nonline), 10))
        c[:, -1] = nonline
        c[:, 0] = 1 - np.array(nonline)
        return c.tolist()

    g5 = []
    rt25 = MS.RuntimeParams(fps=25.0, exp_frame=1, eq_fps=25.0, eq_int_fps=25, exp_time=0.04, runtime_size=[960, 540],
                            raw_size=[1920, 1080], positive_category_list=["METEOR", "RED_SPRITE"])
    rt_exp = MS.RuntimeParams(fps=29.97, exp_frame=3, eq_fps=9.99, eq_int_fps=10, exp_time=0.1, runtime_size=[960, 540],
                              raw_size=[3840, 2160], positive_category_list=["METEOR", "RED_SPRITE"])

    def streak(f0, nframes, p0, v, length, rt, jitter=0, nonline=0.05, extra=None):
        sc = []
        rng = np.random.default_rng(f0 + nframes)
        for k in range(nframes):
            fr = f0 + k * rt.exp_frame
            hx = p0[0] + v[0] * k
            hy = p0[1] + v[1] * k
            norm = (v[0]**2 + v[1]**2)**0.5
            tx = hx - v[0] / norm * length
            ty = hy - v[1] / norm * length
            jit = rng.integers(-jitter, jitter + 1, 4) if jitter else np.zeros(4, int)
            ln = [[int(tx + jit[0]), int(ty + jit[1]), int(hx + jit[2]), int(hy + jit[3])]]
            nl = [nonline]
            if extra:
                ln += extra(k)
                nl += [nonline] * len(extra(k))
            sc.append((fr, ln, m3_cates(nl)))
        return sc

    scripts = {
        "moderate_meteor": ("normal", rt25, streak(40, 14, (300, 200), (6, 3), 28, rt25, jitter=1), 300),
        "moderate_meteor_high": ("high", rt25, streak(75, 20, (600, 400), (-5, 4), 22, rt25, jitter=1, nonline=0.2), 320),
        "single_streak": ("normal", rt25, streak(50, 12, (200, 200), (26, 13), 40, rt25), 300),
        "single_streak_jitter": ("normal", rt25, streak(30, 16, (700, 100), (-14, 9), 30, rt25, jitter=2), 250),
        "slow_long_plane": ("normal", rt25, streak(10, 120, (100, 300), (3, 0.5), 25, rt25), 400),
        "two_simultaneous": ("normal", rt25,
                             sorted(streak(40, 10, (150, 120), (20, 10), 35, rt25) + [
                                 (fr, ln, ct) for fr, ln, ct in streak(43, 9, (800, 400), (-18, -8), 30, rt25)
                             ], key=lambda r: r[0]), 300),
        "exp3_streak": ("high", rt_exp, streak(60, 8, (300, 250), (40, -22), 60, rt_exp), 600),
        "low_preset": ("low", rt25, streak(20, 14, (480, 270), (-22, 12), 45, rt25), 300),
        "area_blob": ("normal", rt25, [(fr, [[400 + (fr % 3), 300, 430, 322 + (fr % 2)]], m3_cates([0.9]))
                                       for fr in range(60, 90)], 300),
        "two_events_far_apart": ("normal", rt25,
                                 streak(20, 10, (200, 200), (25, 5), 40, rt25) +
                                 streak(260, 10, (600, 300), (-20, 15), 40, rt25), 500),
        "static_flicker": ("normal", rt25, [(fr, [[500, 100, 512, 104]], m3_cates([0.1])) for fr in range(10, 200, 7)], 400),
    }
    # merge frames that appear twice (two_simultaneous): concatenate their lines
    for name, (preset, rt, sc, total) in scripts.items():
        merged = {}
        for fr, ln, ct in sc:
            if fr in merged:
                merged[fr] = (merged[fr][0] + ln, merged[fr][1] + ct)
            else:
                merged[fr] = (ln, ct)
        sc = [(fr, ln, ct) for fr, (ln, ct) in sorted(merged.items())]
        recs, rows = run_script(preset, rt, sc, total=total)
        g5.append({"name": name, "preset": preset, "rt": rt.__dict__, "total": total,
                   "script": [[fr, ln, ct] for fr, ln, ct in sc], "records": recs, "log": rows})
    # > 10 lines in one frame: exercises the axis-0 arg-max truncation (collector.py:486-487)
    rng = np.random.default_rng(5150)
    many = []
    for fr in range(30, 42):
        ln = []
        for q in range(14):
            x, y = 100 + 60 * q, 80 + 25 * q + (fr - 30) * 6
            ln.append([x, y, x + 30 + int(rng.integers(0, 5)), y + 12])
        many.append((fr, ln, m3_cates(list(rng.uniform(0, 0.6, 14)))))
    recs, rows = run_script("normal", rt25, many, total=300)
    g5.append({"name": "fourteen_lines_per_frame", "preset": "normal", "rt": rt25.__dict__, "total": 300,
               "script": [[fr, ln, ct] for fr, ln, ct in many], "records": recs, "log": rows})
    json.dump(g5, open(os.path.join(HERE, "g5_collector.json"), "w"))

    # ---------------------------------------------------------------- G10: the MDRF envelope
    import dataclasses
    import typing

    def build(tp, value):
        # what dacite.from_dict does for these dataclasses (dacite itself is absent from this image): nested dataclasses by
        # field, keys the dataclass does not declare are dropped
        if dataclasses.is_dataclass(tp) and isinstance(value, dict):
            hints = typing.get_type_hints(tp)
            return tp(**{f.name: build(hints[f.name], value[f.name]) for f in dataclasses.fields(tp) if f.name in value})
        if typing.get_origin(tp) is typing.Union:
            for arg in typing.get_args(tp):
                if dataclasses.is_dataclass(arg) and isinstance(value, dict):
                    try:
                        return build(arg, value)
                    except TypeError:
                        continue
        return value

    g10 = []
    for preset, script_name in (("normal", "two_simultaneous"), ("high", "exp3_streak"), ("low", "low_preset")):
        cfg_dict = json.load(open(os.path.join(REF, "config", f"m3det_{preset}.json"), encoding="utf-8"))
        cfg = build(MS.MainDetectCfg, cfg_dict)
        p_, rt, sc, total = scripts[script_name]
        merged = {}
        for fr, ln, ct in sc:
            merged[fr] = (merged[fr][0] + ln, merged[fr][1] + ct) if fr in merged else (ln, ct)
        sc = [(fr, ln, ct) for fr, (ln, ct) in sorted(merged.items())]
        log = CaptureLog()
        col = CO.MeteorCollector(collector_cfg(p_), rt, None, log)
        by_frame = {fr: (ln, ct) for fr, ln, ct in sc}
        for i in range(0, total, rt.exp_frame):
            if i in by_frame:
                col.update(i, lines=np.array(by_frame[i][0], dtype=np.int32).reshape(-1, 4), cates=np.array(by_frame[i][1], dtype=np.float64))
            elif (i // rt.exp_frame) % rt.eq_int_fps == 0:
                col.update(i, lines=[], cates=[])
        col.clear()
        info = MS.BasicInfo(loader=cfg.loader.name, video="synthetic.mp4", mask="mask-east.jpg" if preset != "low" else None,
                            start_time=0, end_time=int(total / rt.fps * 1000), resolution=list(rt.raw_size),
                            runtime_resolution=list(rt.runtime_size), exp_time=rt.exp_time, total_frames=total, fps=rt.fps,
                            desc=None if preset == "normal" else f"{preset} preset")
        doc = MS.MDRF(version=U.VERSION, basic_info=info, config=cfg, type="prediction", anno_size=info.resolution,
                      results=col.met_exporter.meteor_list)
        g10.append({"preset": preset, "script": script_name, "config_json": cfg_dict, "basic_info": info.to_dict(),
                    "mdrf": doc.to_dict()})
    json.dump(g10, open(os.path.join(HERE, "g10_mdrf.json"), "w"))

    # ---------------------------------------------------------------- G11: visu() attribute records
    g11 = []
    for seed, (h, w), roi_on in ((1, (27, 48), True), (2, (54, 96), True), (3, (20, 20), False)):
        rng = np.random.default_rng(seed)
        mask = np.ones((h, w), np.uint8)
        det = D.M3Detector(window_sec=1, fps=4, mask=mask, num_cls=U.get_num_class(),
                           cfg=MS.BinaryCfg(MS.BinaryCoreCfg(True, 7, "normal", 0.1, 2), MS.HoughLineCfg(10, 10, 10), MS.DynamicCfg(True, 5)),
                           logger=None)
        state = dict(dst=((rng.random((h, w)) < 0.05) * 255).astype(np.uint8), snr=float(rng.uniform(0.5, 3)), bi_threshold=int(rng.integers(3, 12)),
                     bi_threshold_float=float(rng.uniform(3, 12)), lines_num=int(rng.integers(0, 30)), filtered_line_num=int(rng.integers(0, 5)),
                     dst_sum=float(rng.uniform(0, 0.2)))
        det.dst = state["dst"]
        det.stack.noise_ema.cur_value = state["snr"]
        for k in ("bi_threshold", "bi_threshold_float", "lines_num", "filtered_line_num", "dst_sum"):
            setattr(det, k, state[k])
        if not roi_on:
            det.stack.std_roi = None
        recs = []
        for item in det.visu():
            d = dataclasses.asdict(item)
            if d.get("img") is not None:
                d["img"] = np.asarray(d["img"]).tolist()
            recs.append({"class": type(item).__name__, "fields": d})
        state["dst"] = state["dst"].tolist()
        g11.append({"shape": [h, w], "std_roi": list(det.stack.std_roi) if roi_on else None, "state": state, "visu": recs})
    json.dump(g11, open(os.path.join(HERE, "g11_visu.json"), "w"))

    # ---------------------------------------------------------------- G6
    g6 = {"frame2ts": [[f, fps, U.frame2ts(f, fps)] for f in (0, 1, 24, 25, 1499, 90000, 65536)
                       for fps in (25.0, 29.97, 9.99, 60.0)],
          "ts2frame": [[ts, fps, U.ts2frame(ts, fps)] for ts in ("00:00:02.56", "00:01:00", "01:02:03.004", "00:00:00.020")
                       for fps in (25.0, 29.97)],
          "time2frame": [[ms, fps, U.time2frame(ms, fps)] for ms in (0, 20, 1000, 2400, 123456) for fps in (25.0, 29.97)],
          "frame2time": [[f, fps, U.frame2time(f, fps)] for f in (0, 1, 60, 1499) for fps in (25.0, 29.97)]}
    json.dump(g6, open(os.path.join(HERE, "g6_time.json"), "w"))

    # ---------------------------------------------------------------- G7
    g7 = []
    for (m, w, count) in [(1 - 2 / 60, 8, 40), (1 - 1 / 60, 25, 80), (0.99, 1, 150), (0.9, 0, 10)]:
        e = U.EMA(momentum=m, warmup_speed=w)
        rng = np.random.default_rng(7)
        xs = rng.uniform(1, 4, count)
        vals, moms = [], []
        for x in xs:
            e.update(x)
            vals.append(float(e.cur_value))
            moms.append(float(e.cur_momentum))
        g7.append({"momentum": m, "warmup_speed": w, "x": xs.tolist(), "value": vals, "cur_momentum": moms})
    json.dump(g7, open(os.path.join(HERE, "g7_ema.json"), "w"))
    print("golden fixtures written to", HERE)
    # ---------------------------------------------------------------- G8: exposure estimation
    gen_exposure()


class ScriptedLoader:
    """The attributes and calls rf_estimator / _rf_est_kernel / init_exp_time make on a video loader, over a scripted series
    of per-frame pixel sums (each frame is a 1-pixel image holding its sum: np.sum(frame) is all the reference looks at)."""

    def __init__(self, sums, fps):
        self.all = [float(v) for v in sums]
        self.start_frame, self.end_frame = 0, len(self.all)
        self.exp_frame = 1
        self.pos = 0
        self.windows = []

        class _V:
            pass

        self.video = _V()
        self.video.fps = fps

        class _L:
            def info(self, *a):
                pass

            warning = info

        self.logger = _L()

    @property
    def iterations(self):
        return self.cur_end - self.cur_start if hasattr(self, "cur_start") else self.end_frame - self.start_frame

    def reset(self, start_frame=None, end_frame=None, **kw):
        self.cur_start = self.start_frame if start_frame is None else start_frame
        self.cur_end = self.end_frame if end_frame is None else end_frame

    def start(self):
        if not hasattr(self, "cur_start"):
            self.reset()
        self.pos = self.cur_start
        self.windows.append([self.cur_start, self.cur_end])

    @property
    def stopped(self):
        return self.pos >= self.cur_end

    def pop(self):
        v = self.all[self.pos]
        self.pos += 1
        return np.array([[v]])

    def stop(self):
        pass


def exposure_series(seed, length, period, jitter=0.0, drift=0.0):
    """Per-frame sums of a camera whose real exposure spans `period` video frames: the sum jumps when a new exposure starts and
    repeats (plus noise) in between."""
    rng = np.random.default_rng(seed)
    out, level = [], 5.0e7
    for i in range(length):
        if i % period == 0:
            level = 5.0e7 + rng.normal(0, 4.0e5) + drift * i
        out.append(float(np.rint(level + rng.normal(0, jitter))))
    return out


def gen_exposure():
    from MetLib import videoloader as VL
    from MetLib.utils import sigma_clip
    g8 = {"rf": [], "sigma_clip": [], "init_exp_time": []}
    cases = [(1, 120, 1, 0.0), (2, 250, 2, 50.0), (3, 299, 3, 500.0), (4, 300, 2, 100.0), (5, 1000, 4, 2000.0), (6, 777, 5, 0.0),
             (7, 90, 6, 3.0e4), (8, 400, 1, 1.0e5), (9, 10, 2, 0.0), (10, 4, 1, 0.0)]
    for seed, length, period, jitter in cases:
        sums = exposure_series(seed, length, period, jitter)
        ld = ScriptedLoader(sums, 25.0)
        est = VL.rf_estimator(ld)
        g8["rf"].append(dict(seed=seed, length=length, period=period, jitter=jitter, sums=sums, windows=ld.windows,
                             estimate=float(est)))
    rng = np.random.default_rng(3)
    for k in range(8):
        seq = rng.integers(1, 6, int(rng.integers(3, 40))).tolist()
        if k % 2:
            seq += [int(rng.integers(20, 90))]
        g8["sigma_clip"].append(dict(seq=seq, out=[int(v) for v in sigma_clip(seq)]))
    for opt, fps, ub in [("auto", 25.0, 0.25), ("auto", 50.0, 0.5), ("slow", 30.0, 0.25), ("real-time", 30.0, 0.25),
                         ("0.5", 25.0, 0.25), (0.01, 25.0, 0.25), (0.2, 25.0, 0.25), ("auto", 3.0, 0.25), (1, 4.0, 0.25)]:
        sums = exposure_series(42, 260, 3, 80.0)
        ld = ScriptedLoader(sums, fps)
        val = VL.VanillaVideoLoader.init_exp_time(ld, opt, ub)
        g8["init_exp_time"].append(dict(option=opt, fps=fps, upper_bound=ub, sums=sums, exp_time=float(val),
                                        exp_frame=int(round(float(val) * fps))))
    json.dump(g8, open(os.path.join(HERE, "g8_exposure.json"), "w"))


def gen_config1_truth():
    """config1_truth.json: the DATA the reference holds about its missing sample video test/20220413Red.mp4 (a Git-LFS blob) --
    test/20220413_annotation.json (whole file) and entry 0 of test/clip_test.json -- copied as data; metdetpy_amd.synth.Config1Clip
    builds the stand-in clip of BASELINE config 1 from it (tests/config1_chain.py, tests/test_config1_standin.py)."""
    truth = json.load(open(os.path.join(REF, "test", "20220413_annotation.json"), encoding="utf-8"))
    clip = json.load(open(os.path.join(REF, "test", "clip_test.json"), encoding="utf-8"))[0]
    out = {"source": "data the reference holds about its missing sample video test/20220413Red.mp4 (a Git-LFS blob): "
                     "test/20220413_annotation.json (whole file) and test/clip_test.json entry 0",
           "annotation": truth, "clip_test_entry_0": clip}
    json.dump(out, open(os.path.join(HERE, "config1_truth.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
    gen_config1_truth()
