"""MeteorCollector: turns per-frame segments into meteor records (the bookkeeping surface behind the detector).

`MeteorCollector` has the reference's constructor / `update(frame, lines, cates)` / `clear()` /
`met_exporter.meteor_list` surface (collector.py:372-631) and produces the same records, including the quirks the
reference's output depends on:

  * class ids are taken with an arg-max over the LINE axis and zipped with the lines, so at most `num_cls`
    lines per frame are used (collector.py:486-487);
  * `Name2Label.OTHERS() == 8` / `DROPPED() == 9` while the id -> name table says 8 = "DROPPED", 9 = "OTHERS"
    (collector.py:56-63 vs utils.py:43-46);
  * `clear()` flushes by pretending the current frame is 65536 (collector.py:587).

The per-frame bookkeeping (series association, statistics, scoring) is native code written for this build
(csrc/metdet_collector.hpp behind the `mdh_collector_*` C-ABI): a whole detector pass goes in with one call
(`update_batch`).  Python keeps what happens once per finished series: rounding, time stamps, the merge of an exported
group into records and the rescale (`MetExporter`).  A line-by-line Python restatement of the reference's classes
exists only as the tests' checker, outside this package.

The YOLO re-check of finished clips (collector.py:772-866) is outside this path; `recheck_cfg.switch` must be off.
The exporter works inline instead of on a thread: records come out in the same order.
"""
from __future__ import annotations

import numpy as np

from .metstruct import CollectorCfg, MDTarget, RuntimeParams, SingleMDRecord
from .utils import ID2NAME, NAME2ID, frame2ts, get_num_class


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


# The reference's name is the product: MetDetPy.py:148 constructs `MeteorCollector(...)`.
MeteorCollector = NativeMeteorCollector
