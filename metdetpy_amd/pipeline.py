"""Stream driver: what `MetDetPy.detect_video`'s main loop does between the loader and the MDRF result
(MetDetPy.py:183-250), for frames that are already grayscale at runtime size: detector (batched native passes) ->
MeteorCollector -> list of SingleMDRecord.  Decoding, exposure estimation, the visualiser and the YOLO re-check are
outside this path."""
from __future__ import annotations

from typing import Iterable, Optional

import numpy as np

from .collector import MeteorCollector
from .detector import M3Detector
from .metstruct import MDRF, BasicInfo, BinaryCfg, CollectorCfg, MainDetectCfg, RuntimeParams
from .utils import VERSION, get_num_class


def detect_stream(frames: Iterable[np.ndarray], mask: np.ndarray, detector_cfg: BinaryCfg, collector_cfg: CollectorCfg,
                  rt: RuntimeParams, *, window_sec: float = 1.0, batch: int = 64, device: int = 0, logger=None,
                  detector_cls=M3Detector, start_frame: int = 0, collector_cls=MeteorCollector, detector_kwargs=None):
    """Feed `frames` (each (H,W) uint8, one per exposure frame) through the detector and the collector exactly in the
    order the reference's loop does: collector.update(i, lines, cates) when the frame has lines, or on every
    eq_int_fps-th iteration (MetDetPy.py:200-202); collector.clear() at the end (MetDetPy.py:237).
    The result arrays of a pass go to the (native) collector in one call.  `collector_cls` exists for the tests, which run their
    Python restatement of the reference's collector through the same loop frame by frame.  Returns (records, per_frame_info)."""
    det = detector_cls(window_sec, rt.eq_fps, mask, get_num_class(), detector_cfg, logger, device=device, max_batch=batch,
                       **(detector_kwargs or {}))
    col = collector_cls(collector_cfg, rt, None, logger)
    native_collector = hasattr(col, "update_batch")
    infos = []
    it = 0
    buf = []

    def flush():
        nonlocal it
        if not buf:
            return
        if native_collector and detector_cls is M3Detector:
            info, raw, nms, prob = det._ctx.process_batch(np.stack(buf))
            col.update_batch(start_frame + it * rt.exp_frame, it, info, nms, prob)
            # the detector's public attributes follow the stream as they do in the reference's loop (bi_threshold, stack.snr,
            # lines_num, dst_sum, linesp_ext: what visu() and a caller looking at the detector mid-stream read)
            det._absorb_info(info[-1])
            det.linesp_ext = det._package(info[-1], raw[-1], nms[-1], prob[-1])[0]
            it += len(buf)
            infos.append(info)
            buf.clear()
            return
        results, info = det.detect_batch(np.stack(buf))
        for (lines, cates) in results:
            i = start_frame + it * rt.exp_frame
            if len(lines) or (it % rt.eq_int_fps == 0):
                col.update(i, lines=lines, cates=cates)
            it += 1
        infos.append(info)
        buf.clear()

    for f in frames:
        buf.append(f)
        if len(buf) == batch:
            flush()
    flush()
    col.clear()
    det.close()
    return col.met_exporter.meteor_list, (np.concatenate(infos) if infos else np.zeros(0))


def to_mdrf(records, basic_info: BasicInfo, cfg: Optional[MainDetectCfg] = None) -> MDRF:
    """What `detect_video` returns (MetDetPy.py:245-250): `to_mdrf(...).to_dict()` is the document MetDetPy.py:395-398 writes,
    key for key (version / basic_info / config / type / anno_size / results / performance)."""
    return MDRF(version=VERSION, basic_info=basic_info, config=cfg, type="prediction", anno_size=basic_info.resolution,
                results=list(records))


def calculate_time_iou(a, b) -> float:
    """evaluate.py:119-139 on [start_frame, last_activate_frame]."""
    if a.start_frame >= b.last_activate_frame or a.last_activate_frame <= b.start_frame:
        return 0
    t = sorted([a.start_frame, a.last_activate_frame, b.start_frame, b.last_activate_frame], reverse=True)
    return (t[1] - t[2]) / (t[0] - t[3])


def calculate_area_iou(b1, b2) -> float:
    """utils.py:921-953 on Box(x1,y1,x2,y2)."""
    if b1.x1 == b1.x2 == b2.x1 == b2.x2:
        oy = sorted([b1.y1, b1.y2, b2.y1, b2.y2], reverse=True)
        if b1.y1 == b2.y1 and b1.y2 == b2.y2:
            return 1
        return (oy[2] - oy[1]) / (oy[3] - oy[0])
    if b1.y1 == b1.y2 == b2.y1 == b2.y2:
        ox = sorted([b1.x1, b1.x2, b2.x1, b2.x2], reverse=True)
        if b1.x1 == b2.x1 and b1.x2 == b2.x2:
            return 1
        return (ox[2] - ox[1]) / (ox[3] - ox[0])
    if b1.x1 >= b2.x2 or b1.x2 <= b2.x1 or b1.y1 >= b2.y2 or b1.y2 <= b2.y1:
        return 0
    ix = sorted([b1.x1, b1.x2, b2.x1, b2.x2], reverse=True)[1:-1]
    iy = sorted([b1.y1, b1.y2, b2.y1, b2.y2], reverse=True)[1:-1]
    inter = (ix[1] - ix[0]) * (iy[1] - iy[0])
    return inter / ((b1.x2 - b1.x1) * (b1.y2 - b1.y1) + (b2.x2 - b2.x1) * (b2.y2 - b2.y1) - inter)
