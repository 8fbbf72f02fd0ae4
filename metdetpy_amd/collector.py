"""MeteorCollector: turns per-frame segments into meteor records (the bookkeeping surface behind the detector).

Host-side Python, same constructor / `update(frame, lines, cates)` / `clear()` / `met_exporter.meteor_list`
surface and the same record contents as the reference (collector.py:28-770), including the quirks its output
depends on:

  * class ids are taken with an arg-max over the LINE axis and zipped with the lines, so at most `num_cls`
    lines per frame are used (collector.py:486-487);
  * `Name2Label.OTHERS() == 8` / `DROPPED() == 9` while the id -> name table says 8 = "DROPPED", 9 = "OTHERS"
    (collector.py:56-63 vs utils.py:43-46);
  * `clear()` flushes by pretending the current frame is 65536 (collector.py:587).

The YOLO re-check of finished clips (collector.py:772-866) is outside this path; `recheck_cfg.switch` must be off.
The exporter works inline instead of on a thread: records come out in the same order.
"""
from __future__ import annotations

import numpy as np

from .metstruct import CollectorCfg, MDTarget, RuntimeParams, SingleMDRecord
from .utils import ID2NAME, NAME2ID, frame2ts, get_num_class, pt_drct, pt_len, pt_len_sqr


class Name2Label:
    METEOR = 0
    PLANE_SATELLITE = 1
    RED_SPRITE = 2
    LIGHTNING = 3
    JET = 4
    RARE_SPRITE = 5
    SPACECRAFT = 6
    BUGS = 7

    @staticmethod
    def OTHERS():
        return get_num_class() - 2

    @staticmethod
    def DROPPED():
        return get_num_class() - 1


def scale_to(pt, ratio):
    return [int(v * r) for v, r in zip(pt, ratio)]


def create_prob_func(rng):
    """Trapezoid x/a | 1 | (2b-x)/b | 0 over [0,a) [a,b] (b,2b) [2b,inf)  (collector.py:70-94)."""
    a, b = rng

    def prob(x):
        if x < a:
            return x / a
        if a <= x <= b:
            return 1
        if x < 2 * b:
            return (2 * b - x) / b
        return 0

    return prob


class PointList:
    """Growing (k,2) point array with the frame each point was seen in (collector.py:97-138)."""

    def __init__(self):
        self.pts = np.zeros((0, 2), dtype=np.int32)
        self.frame_num = np.zeros((0, ), dtype=np.int16)

    def extend(self, new_pts, frame):
        self.pts = np.concatenate([self.pts, np.array(new_pts)], axis=0)
        self.frame_num = np.concatenate([self.frame_num, np.ones((len(new_pts), )) * frame], axis=0)

    def get_pts_as_list(self):
        return [[int(p[0]), int(p[1])] for p in self.pts]

    def __getitem__(self, i):
        return self.pts[i]

    def __len__(self):
        return len(self.pts)


class MeteorSeries:
    """One candidate event: the responses that were chained together (collector.py:141-369)."""

    def __init__(self, start_frame, cur_frame, init_pts, max_acceptable_dist, max_acti_frame, cate_prob, fps,
                 runtime_size):
        assert len(init_pts) in (3, 5), f"invalid init_pts length: should be 3 but {len(init_pts)} got."
        self.coord_list = PointList()
        self.center_list = PointList()
        self.drct_list = []
        self.coord_list.extend(init_pts, cur_frame)
        self.center_list.extend(np.mean(init_pts, axis=0)[None, :], cur_frame)
        with np.errstate(invalid="ignore", divide="ignore"):
            self.drct_list.append(pt_drct(init_pts[0], init_pts[1]))
        self.start_frame = start_frame
        self.end_frame = cur_frame
        self.last_activate_frame = cur_frame
        self.max_acti_frame = max_acti_frame
        self.max_acceptable_dist = max_acceptable_dist
        self.count = 1
        self.cate_prob = cate_prob
        self.fps = fps
        self.runtime_length = max(runtime_size)
        self.range = ([2**16, 2**16], [-2**16, -2**16])
        self.calc_new_range(init_pts)

    @property
    def drst_std(self):
        if len(self.drct_list) == 0:
            return 0
        d = np.array(self.drct_list.copy())
        s1 = np.std(np.sort(d)[:-1]) if len(d) >= 3 else np.std(d)
        d[d > np.pi / 2] -= np.pi
        s2 = np.std(np.sort(d)[:-1]) if len(d) >= 3 else np.std(d)
        return min(s1, s2)

    @property
    def cate(self):
        return np.argmax(self.cate_prob, axis=0)

    @property
    def duration(self):
        return self.last_activate_frame - self.start_frame + 1

    @property
    def fix_duration(self):
        return self.duration / self.fps

    @property
    def fix_motion_duration(self):
        return (self.end_frame - self.start_frame) / self.fps

    @property
    def sort_range(self):
        [x0, y0], [x1, y1] = self.range
        e_x, e_y = self.coord_list[int(np.argmin(self.coord_list.frame_num))]
        l_x, l_y = self.coord_list[int(np.argmax(self.coord_list.frame_num))]
        if e_x > l_x:
            x0, x1 = x1, x0
        if e_y > l_y:
            y0, y1 = y1, y0
        return [x0, y0], [x1, y1]

    @property
    def dist(self):
        return pt_len(*self.range)

    @property
    def fix_dist(self):
        return self.dist / self.runtime_length * 100

    @property
    def speed(self):
        return self.dist / (self.end_frame - self.start_frame + 1e-6)

    @property
    def fix_speed(self):
        return self.speed * self.fps / self.runtime_length * 100

    def get_met_attr(self, decimals: int = 3) -> MDTarget:
        pt1, pt2 = self.sort_range
        return MDTarget(start_time=frame2ts(self.start_frame, self.fps), start_frame=self.start_frame,
                        end_time=frame2ts(self.end_frame, self.fps), last_activate_frame=self.last_activate_frame,
                        last_activate_time=frame2ts(self.last_activate_frame, self.fps), duration=self.duration,
                        speed=np.round(self.speed, decimals), dist=np.round(pt_len(pt1, pt2), decimals),
                        fix_dist=np.round(self.fix_dist, decimals), fix_speed=np.round(self.fix_speed, decimals),
                        fix_motion_duration=np.round(self.fix_motion_duration, decimals),
                        fix_duration=np.round(self.fix_duration, decimals), num_pts=len(self.coord_list),
                        category=ID2NAME[self.cate], pt1=pt1, pt2=pt2,
                        center_point_list=self.center_list.get_pts_as_list(), drct_loss=np.round(self.drst_std, 3),
                        score=-1, real_dist=-1)

    def calc_new_range(self, pts):
        lo, hi = self.range
        self.range = [min(int(min(p[0] for p in pts)), lo[0]),
                      min(int(min(p[1] for p in pts)), lo[1])], [max(int(max(p[0] for p in pts)), hi[0]),
                                                                 max(int(max(p[1] for p in pts)), hi[1])]

    def update(self, new_frame, new_box, new_cate):
        (x1, y1), (x2, y2) = self.range
        assert len(new_box) in (3, 5), f"invalid init_pts length: should be 3 but {len(new_box)} got."
        for pt in new_box:  # a point outside the span so far means the event is still moving
            if not ((x1 <= pt[0] <= x2) and (y1 <= pt[1] <= y2)):
                self.end_frame = new_frame
                break
        self.last_activate_frame = new_frame
        self.coord_list.extend(new_box, new_frame)
        self.center_list.extend(np.mean(new_box, axis=0)[None, :], new_frame)
        self.calc_new_range(new_box)
        with np.errstate(invalid="ignore", divide="ignore"):
            self.drct_list.append(pt_drct(new_box[0], new_box[1]))
        self.cate_prob += new_cate
        self.count += 1

    def may_in_series(self, pts, cur_frame):
        """True when any new point lies within sqrt(max_acceptable_dist) of a point seen in the last max_acti_frame
        frames (collector.py:357-369).  Same integer comparisons as the reference's double loop, vectorised."""
        recent = np.where(self.coord_list.frame_num >= cur_frame - self.max_acti_frame)[0]
        if len(recent) == 0:
            return False
        old = self.coord_list.pts[recent[0]:]
        new = np.asarray(pts)
        d2 = (new[:, None, 1] - old[None, :, 1])**2 + (new[:, None, 0] - old[None, :, 0])**2
        return bool(np.any(d2 < self.max_acceptable_dist))


class _Joinable:

    def join(self, timeout=None):
        return None


class MetExporter:
    """Merges finished series into records and rescales them to the raw resolution (collector.py:634-770)."""
    END_FLAG = "END_FLAG"
    DROP_FLAG = "DROP_FLAG"
    ACTIVE_FLAG = "ACTIVE_FLAG"

    def __init__(self, recheck_cfg, runtime_param: RuntimeParams, video_loader, logger, max_interval, det_thre):
        self.recheck = bool(getattr(recheck_cfg, "switch", False))
        if self.recheck:
            raise NotImplementedError("the YOLO re-check is outside the MI355X hot path: run with recheck off "
                                      "(MetDetPy.py --recheck off / recheck_cfg.switch = false)")
        self.positive_cates = runtime_param.positive_category_list
        self.positive_cate_ids = [NAME2ID[c] for c in self.positive_cates if c in NAME2ID]
        self.logger = logger
        self.max_interval = max_interval
        self.det_thre = det_thre
        self.fps = runtime_param.fps
        self.raw_size = runtime_param.raw_size
        self.rescale_ratio = [x / y for x, y in zip(runtime_param.raw_size, runtime_param.runtime_size)]
        self.export_loop = _Joinable()
        self.meteor_list = []
        self.closed = False

    def _log(self, kind, text):
        fn = getattr(self.logger, kind, None) if self.logger is not None else None
        if fn is not None:
            fn(text)

    def export(self, flag, data):
        if self.closed:
            return
        if flag == self.DROP_FLAG:
            for tgt in data:
                tgt.category = ID2NAME[Name2Label.DROPPED()]
                rec = self.rescale(SingleMDRecord.from_target(tgt, self.raw_size))
                self.meteor_list.append(rec)
                self._log("dropped", rec.to_json(full=False))
        elif flag == self.ACTIVE_FLAG:
            cur, finished = None, []
            for tgt in data:
                if cur is None:
                    cur = SingleMDRecord.from_target(tgt, self.raw_size)
                    continue
                if tgt.start_frame < cur.end_frame + self.max_interval:
                    if tgt.last_activate_frame > cur.end_frame:
                        cur.end_frame = tgt.last_activate_frame
                        cur.end_time = tgt.last_activate_time
                    cur.target.append(tgt)
                else:
                    finished.append(cur)
                    cur = SingleMDRecord.from_target(tgt, self.raw_size)
            if cur is not None:
                finished.append(cur)
            for rec in finished:
                rec = self.rescale(rec)
                self.meteor_list.append(rec)
                self._log("meteor", rec.to_json(full=False))
        elif flag == self.END_FLAG:
            self.closed = True
        else:
            raise KeyError(f"Unexpected flag received. Except [{self.ACTIVE_FLAG}{self.DROP_FLAG},{self.END_FLAG}], "
                           f"got {flag} instead.")

    def rescale(self, rec: SingleMDRecord) -> SingleMDRecord:
        for t in rec.target:
            t.pt1 = scale_to(t.pt1, self.rescale_ratio)
            t.pt2 = scale_to(t.pt2, self.rescale_ratio)
            t.real_dist = t.dist * max(self.rescale_ratio)
            t.center_point_list = [scale_to(p, self.rescale_ratio) for p in t.center_point_list]
        return rec


class MeteorCollector:
    """collector.py:372-631."""

    def __init__(self, collector_cfg: CollectorCfg, runtime_param: RuntimeParams, video_loader=None, logger=None):
        m = collector_cfg.meteor_cfg
        self.min_len = m.min_len
        self.max_interval = m.max_interval * runtime_param.fps
        self.max_acti_frame = int(m.max_interval * runtime_param.fps)
        self.det_thre = m.det_thre
        self.thre2 = m.thre2 * runtime_param.exp_frame
        self.runtime_size = runtime_param.runtime_size
        # sentinel that never retires; new series are inserted in front of it
        self.active_meteor = [
            MeteorSeries(2**16, 2**16, np.array([[-100, -100], [-101, -101], [-102, -102]]), np.nan, np.nan, None,
                         runtime_param.fps, self.runtime_size)
        ]
        self.waiting_meteor = []
        self.cur_frame = 0
        self.eframe = runtime_param.exp_frame
        self.fps = runtime_param.fps
        self.time_prob_func = create_prob_func(m.time_range)
        self.speed_prob_func = create_prob_func(m.speed_range)
        self.len_prob_func = create_prob_func((self.min_len, np.inf))
        self.drct_prob_func = create_prob_func(m.drct_range)
        self.logger = logger
        self.met_exporter = MetExporter(collector_cfg.recheck_cfg, runtime_param, video_loader=video_loader, logger=logger,
                                        max_interval=self.max_interval, det_thre=self.det_thre)

    def update(self, cur_frame, lines, cates):
        self.cur_frame = cur_frame
        keep, drop = [], []
        for ms in self.active_meteor:
            if self.cur_frame - ms.last_activate_frame >= self.max_interval:
                if (self.prob_meteor(ms) > self.det_thre / 2) and (self.prob_meteor(ms) != self.det_thre):
                    if self.met_exporter.recheck or not (ms.cate in [Name2Label.OTHERS(), Name2Label.PLANE_SATELLITE]):
                        keep.append(ms)
                    else:
                        drop.append(ms)
                else:
                    drop.append(ms)
        for ms in drop:
            self.active_meteor.remove(ms)
        for ms in keep:
            self.active_meteor.remove(ms)
        self.met_exporter.export(self.met_exporter.DROP_FLAG, [self.get_met_attr(ms) for ms in drop])
        self.waiting_meteor.extend(keep)
        if len(self.waiting_meteor) > 0:
            quiet = True
            for ms in self.active_meteor:
                if self.prob_meteor(ms) > self.det_thre / 2 and \
                        (ms.start_frame - self.waiting_meteor[-1].last_activate_frame <= self.max_interval):
                    quiet = False
                    break
            if quiet:
                out = [self.get_met_attr(ms) for ms in self.waiting_meteor]
                out.sort(key=lambda t: t.start_frame)
                self.met_exporter.export(self.met_exporter.ACTIVE_FLAG, out)
                self.waiting_meteor.clear()
        if len(cates) == 0:
            return
        num_active = len(self.active_meteor)
        cate_ids = np.argmax(np.array(cates), axis=0)  # over the LINE axis, as the reference does
        for line_pts, cate_id, cate_prob in zip(lines, cate_ids, cates):
            if cate_id in [Name2Label.METEOR, Name2Label.PLANE_SATELLITE]:
                pts = np.array([line_pts[:2], line_pts[2:], (line_pts[:2] + line_pts[2:]) // 2])
            else:
                x1, y1, x2, y2 = line_pts
                pts = np.array([[x1, y1], [x2, y2], [x2, y1], [x1, y2], [int((x1 + x2) / 2), int((y1 + y2) / 2)]])
            joined = False
            for ms in self.active_meteor[:num_active]:
                if ms.may_in_series(pts, cur_frame):
                    ms.update(self.cur_frame, pts, new_cate=cate_prob)
                    joined = True
                    break
            if joined:
                continue
            self.active_meteor.insert(
                len(self.active_meteor) - 1,
                MeteorSeries(max(self.cur_frame - 2 * self.eframe, 0), self.cur_frame, pts, max_acceptable_dist=self.thre2,
                             max_acti_frame=self.max_acti_frame, cate_prob=cate_prob, fps=self.fps,
                             runtime_size=self.runtime_size))

    def visu(self, frame_num):
        """Name / geometry of what the reference draws (collector.py:530-575); rendering is out of scope."""
        items = []
        for ms in self.active_meteor:
            items.append(dict(name="active_meteors", dot_pair=ms.range, score=self.prob_meteor(ms), category=ID2NAME[ms.cate]))
        return items

    def clear(self):
        self.update(2**16, [], [])
        self.met_exporter.export(self.met_exporter.END_FLAG, [])
        self.met_exporter.export_loop.join()

    def prob_meteor(self, met: MeteorSeries):
        if met.cate == 0:
            return (self.time_prob_func(met.fix_duration) * self.speed_prob_func(met.fix_speed) *
                    self.len_prob_func(met.dist) * self.drct_prob_func(met.drst_std))
        if np.any(np.isnan(met.cate_prob)):
            if self.logger is not None:
                self.logger.error(f"nan detected in cate_prob: {met.cate_prob}")
            raise SystemExit(1)
        return met.cate_prob[met.cate] / met.count

    def get_met_attr(self, met: MeteorSeries) -> MDTarget:
        tgt = met.get_met_attr()
        tgt.score = np.round(self.prob_meteor(met), 2)
        return tgt

    def frame2ts(self, frame):
        return frame2ts(frame, self.fps)


class NativeMeteorCollector:
    """MeteorCollector with the per-frame bookkeeping in the native library (`mdh_collector_*`, csrc/metdet_collector.hpp):
    same constructor, `update` / `clear` / `met_exporter.meteor_list`, same records, plus `update_batch`, which takes the
    result arrays of a whole detector pass in one call.  What stays in Python is what happens once per finished series:
    rounding, time stamps, the merge of an exported group into records and the rescale (MetExporter, as above)."""

    def __init__(self, collector_cfg: CollectorCfg, runtime_param: RuntimeParams, video_loader=None, logger=None):
        import ctypes as C

        from . import _lib
        m = collector_cfg.meteor_cfg
        self._lib, self._C = _lib, C
        self.L = _lib.load()
        c = _lib.MdhCollectorCfg()
        c.min_len, c.max_interval, c.det_thre, c.thre2 = float(m.min_len), float(m.max_interval), float(m.det_thre), float(m.thre2)
        for k in range(2):
            c.time_range[k], c.speed_range[k], c.drct_range[k] = float(m.time_range[k]), float(m.speed_range[k]), float(m.drct_range[k])
        c.fps, c.exp_frame = float(runtime_param.fps), int(runtime_param.exp_frame)
        c.runtime_w, c.runtime_h = int(runtime_param.runtime_size[0]), int(runtime_param.runtime_size[1])
        c.num_cls = get_num_class()
        self._h = self.L.mdh_collector_create(C.byref(c))
        if not self._h:
            raise _lib.MetDetHipError(f"mdh_collector_create failed: {_lib.last_error()}")
        self.fps = runtime_param.fps
        self.eq_int_fps = runtime_param.eq_int_fps
        self.exp_frame = runtime_param.exp_frame
        self.logger = logger
        self.cur_frame = 0
        self.met_exporter = MetExporter(collector_cfg.recheck_cfg, runtime_param, video_loader=video_loader, logger=logger,
                                        max_interval=m.max_interval * runtime_param.fps, det_thre=m.det_thre)

    def _target(self, e, centers) -> MDTarget:
        c = centers[e["center_off"]:e["center_off"] + e["n_center"]]
        return MDTarget(start_time=frame2ts(int(e["start_frame"]), self.fps), start_frame=int(e["start_frame"]),
                        end_time=frame2ts(int(e["end_frame"]), self.fps), last_activate_frame=int(e["last_activate_frame"]),
                        last_activate_time=frame2ts(int(e["last_activate_frame"]), self.fps), duration=int(e["duration"]),
                        speed=np.round(e["speed"], 3), dist=np.round(e["dist"], 3), fix_dist=np.round(e["fix_dist"], 3),
                        fix_speed=np.round(e["fix_speed"], 3), fix_motion_duration=np.round(e["fix_motion_duration"], 3),
                        fix_duration=np.round(e["fix_duration"], 3), num_pts=int(e["num_pts"]), category=ID2NAME[int(e["category"])],
                        pt1=[int(e["pt1"][0]), int(e["pt1"][1])], pt2=[int(e["pt2"][0]), int(e["pt2"][1])],
                        center_point_list=[[int(p[0]), int(p[1])] for p in c], drct_loss=np.round(e["drct_loss"], 3),
                        score=np.round(int(e["score"]) if e["score_is_int"] else e["score"], 2), real_dist=-1)

    def _drain(self):
        C = self._C
        nc = C.c_int(0)
        n = self.L.mdh_collector_events(self._h, None, 0, None, 0, C.byref(nc))
        if n <= 0:
            return
        ev = np.zeros(n, self._lib.TARGET_DTYPE)
        centers = np.zeros((max(nc.value, 1), 2), np.float64)
        got = self.L.mdh_collector_events(self._h, ev.ctypes.data, n, centers.ctypes.data, nc.value, C.byref(nc))
        assert got == n
        start = 0
        for i in range(1, n + 1):  # one export() per group, in order
            if i == n or ev[i]["group"] != ev[start]["group"]:
                flag = self.met_exporter.DROP_FLAG if ev[start]["flag"] == 0 else self.met_exporter.ACTIVE_FLAG
                self.met_exporter.export(flag, [self._target(ev[j], centers) for j in range(start, i)])
                start = i

    def update(self, cur_frame, lines, cates):
        self.cur_frame = cur_frame
        n = len(cates)
        ln = np.ascontiguousarray(np.asarray(lines, dtype=np.int32).reshape(-1, 4)[:max(n, 0)]) if n else np.zeros((0, 4), np.int32)
        ct = np.ascontiguousarray(np.asarray(cates, dtype=np.float64).reshape(n, -1)) if n else np.zeros((0, get_num_class()))
        if n and ct.shape[1] != get_num_class():
            raise ValueError(f"cates must have {get_num_class()} columns")
        self._lib.check(self.L.mdh_collector_update(self._h, int(cur_frame), ln.ctypes.data, ct.ctypes.data, int(n)),
                        "mdh_collector_update")
        self._drain()

    def update_batch(self, frame0: int, it0: int, info, nms, prob):
        """K frames of one detector pass (arrays as Context.collect_into / process_batch return them), gated like the
        reference's main loop (MetDetPy.py:200-202)."""
        K = len(info)
        self._lib.check(self.L.mdh_collector_update_batch(self._h, int(K), int(frame0), int(it0), int(self.eq_int_fps),
                                                          info.ctypes.data, nms.ctypes.data, prob.ctypes.data),
                        "mdh_collector_update_batch")
        self.cur_frame = frame0 + (K - 1) * self.exp_frame if K else self.cur_frame
        self._drain()

    @property
    def num_active(self) -> int:
        return self.L.mdh_collector_active(self._h)

    def clear(self):
        self.update(2**16, [], [])
        self.met_exporter.export(self.met_exporter.END_FLAG, [])
        self.met_exporter.export_loop.join()

    def close(self):
        if getattr(self, "_h", None):
            self.L.mdh_collector_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
