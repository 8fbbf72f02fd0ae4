"""CPU restatement of the reference's MeteorCollector bookkeeping (collector.py:70-632): PointList, MeteorSeries,
MeteorCollector, create_prob_func -- member for member as the reference writes them, including the quirks its output
depends on.

TEST INFRASTRUCTURE ONLY: the checker of the product's native collector (metdetpy_amd/csrc/metdet_collector.hpp behind
`metdetpy_amd.collector.MeteorCollector`).  Imported by tests/, bench.py's parity leg and tools/bench_collector.py; never by
the product package.  Pinned to the reference itself by tests/golden/g5_collector.json (12 scripted runs captured from the
reference's own classes, records and log lines string-equal).

The record / export side (SingleMDRecord, MDTarget, MetExporter: merge of finished series into records, rescale) is shared
with the product: it is schema, and g5 pins it through both collectors.
"""
from __future__ import annotations

import numpy as np

from metdetpy_amd.collector import MetExporter, Name2Label
from metdetpy_amd.metstruct import CollectorCfg, MDTarget, RuntimeParams
from metdetpy_amd.utils import ID2NAME, frame2ts, get_num_class, pt_drct, pt_len, pt_len_sqr


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
