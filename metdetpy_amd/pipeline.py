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


def parse_resize_param(tgt_wh, raw_wh):
    """utils.py:562-617: the loader's `resize` option ([w, h], "WxH", or one integer = the long side) -> [w, h]."""
    if tgt_wh is None:
        return list(raw_wh)
    w, h = raw_wh
    if isinstance(tgt_wh, str):
        try:
            tgt_wh = list(map(int, tgt_wh.lower().split("x"))) if "x" in tgt_wh.lower() else int(tgt_wh)
        except Exception as e:
            raise Exception(f"{e}: unexpected values for argument \"--resize\": input should be either one integer or two integers "
                            f"separated by \"x\", got {tgt_wh}.")
    if isinstance(tgt_wh, int):
        tgt_wh = [tgt_wh, -1] if w > h else [-1, tgt_wh]
    if isinstance(tgt_wh, list):
        if len(tgt_wh) != 2:
            raise Exception(f"Expected tgt_wh is converted to a list with 2 elements, got {len(tgt_wh)}.")
        tgt_wh = list(tgt_wh)
        if tgt_wh[0] <= 0 or tgt_wh[1] <= 0:
            if tgt_wh[0] <= 0 and tgt_wh[1] <= 0:
                return list(raw_wh)
            idn = 0 if tgt_wh[0] <= 0 else 1
            idx = 1 - idn
            tgt_wh[idn] = int(raw_wh[idn] * tgt_wh[idx] / raw_wh[idx])
        return tgt_wh
    raise TypeError(f"Unsupported arg type: it should be <int,str,list>, got {type(tgt_wh)}.")


def detect_clip(read, total_frames: int, fps: float, src_hw, cfg: MainDetectCfg, mask01: np.ndarray, *, device: int = 0,
                batch: int = 64, video: str = "clip", mask_name: Optional[str] = None, logger=None):
    """What `MetDetPy.detect_video` does between the decoder and the MDRF document (MetDetPy.py:24-250), for a clip whose decoded
    frames `read(lo, hi) -> (hi - lo, H, W, 3) uint8 BGR` hands over (decode itself is out of scope):

      exposure estimation on the per-frame sums (GPU) -> exp_frame / eq_fps   videoloader.py:440-494, 796-870
      resize -> BGR2GRAY -> mask -> max over exp_frame source frames (GPU)    imgproc.py:82-139, utils.py:202-204
      n = int(window_sec * eq_fps); detector passes (GPU)                     MetDetPy.py:137-142, Detector.py:186-392
      MeteorCollector (native), collector.clear()                             MetDetPy.py:200-202, 237
      MDRF(version, basic_info, config, "prediction", anno_size = raw size)   MetDetPy.py:245-250

    The re-check (YOLO) is out of scope: the collector runs with recheck_cfg.switch = False whatever the preset says (it refuses True).
    Returns (MDRF, RuntimeParams, per-frame info array)."""
    import copy

    from . import exposure as EX
    from .detector import get_detector
    from .imgproc import Preprocessor
    SH, SW = int(src_hw[0]), int(src_hw[1])
    W, H = parse_resize_param(cfg.loader.resize, [SW, SH])
    if mask01.shape != (H, W):
        raise ValueError(f"mask must be {(H, W)} (the runtime size), got {mask01.shape}")
    chunk = max(1, min(16, total_frames))
    sums_of = lambda lo, hi: np.concatenate([EX.frame_sums(read(a, min(hi, a + chunk)), device=device) for a in range(lo, hi, chunk)])
    exp_time = EX.init_exp_time(cfg.loader.exp_time, fps, cfg.loader.upper_bound, lambda: EX.rf_estimator(sums_of, 0, total_frames), logger)
    ep = EX.exposure_params(float(exp_time), fps)
    e = ep["exp_frame"]
    rt = RuntimeParams(fps=float(fps), exp_frame=e, eq_fps=float(ep["eq_fps"]), eq_int_fps=int(ep["eq_int_fps"]), exp_time=float(exp_time),
                       runtime_size=[W, H], raw_size=[SW, SH], positive_category_list=list(cfg.collector.positive_cfg["positive_cates"]))
    ccfg = copy.deepcopy(cfg.collector)
    ccfg.recheck_cfg.switch = False
    det = get_detector(cfg.detector.name)(cfg.detector.window_sec, rt.eq_fps, mask01, get_num_class(), cfg.detector.cfg, logger,
                                          device=device, max_batch=batch)
    col = MeteorCollector(ccfg, rt, None, logger)
    pre = Preprocessor((SH, SW), (W, H), mask01, device=device)
    iters = total_frames // e  # (a last, incomplete exposure is not handed out: videoloader.py:298-308 pops whole exposures)
    infos, it = [], 0
    while it < iters:
        k = min(batch, iters - it)
        src = np.concatenate([read(a, min((it + k) * e, a + chunk * e)) for a in range(it * e, (it + k) * e, chunk * e)])
        gray = pre.batch(src, e)
        info, _raw, nms, prob = det._ctx.process_batch(gray)
        col.update_batch(it * e, it, info, nms, prob)
        infos.append(info)
        it += k
    col.clear()
    pre.close()
    det.close()
    info_blk = BasicInfo(loader=cfg.loader.name, video=video, mask=mask_name, start_time=0, end_time=int(total_frames / fps * 1000),
                         resolution=[SW, SH], runtime_resolution=[W, H], exp_time=rt.exp_time, total_frames=total_frames, fps=float(fps))
    return to_mdrf(col.met_exporter.meteor_list, info_blk, cfg), rt, (np.concatenate(infos) if infos else np.zeros(0))


def match_annotation(mdrf: MDRF, annotation: dict, fps: float, real_size_wh, pos_thre: float = 0.5, tiou: float = 0.3, aiou: float = 0.3):
    """evaluate.py's matching rule (`compare`, evaluate.py:181-275) between a prediction and a ground-truth annotation in the form
    the reference's test/20220413_annotation.json has ({"anno_size": [w, h], "meteors": [{"start_time", "end_time", "pt1", "pt2"}]}):
    both sides' end points are scaled from their anno_size to the video's real size with int() (`scale`, evaluate.py:22-23,
    100-116), predictions with score <= pos_thre are skipped (gt_mode), and a prediction matches a not-yet-matched annotation when
    the time IoU on [start_frame, last_activate_frame] (evaluate.py:119-139) is >= tiou AND the box IoU (utils.py:921-953) is
    >= aiou.  Annotation times become frames at the video's fps.  Returns [(prediction MDTarget, annotation index, tIoU, IoU)]
    and the indices of the annotations left unmatched."""
    from .metstruct import Box
    from .utils import ts2frame
    dx, dy = real_size_wh

    def scaled(pt, size):
        return [int(pt[0] * (dx / size[0])), int(pt[1] * (dy / size[1]))]

    class _Gt:
        def __init__(self, m):
            self.start_frame = ts2frame(m["start_time"][:12], fps)
            self.last_activate_frame = ts2frame(m["end_time"][:12], fps)
            self.box = Box.from_pts(scaled(m["pt1"], annotation["anno_size"]), scaled(m["pt2"], annotation["anno_size"]))

    gts = [_Gt(m) for m in annotation["meteors"]]
    taken = [False] * len(gts)
    pairs = []
    for rec in mdrf.results:
        for t in rec.target:
            if t.score <= pos_thre:
                continue
            box = Box.from_pts(scaled(t.pt1, mdrf.anno_size), scaled(t.pt2, mdrf.anno_size))
            for j, g in enumerate(gts):
                if taken[j]:
                    continue
                ti, ai = calculate_time_iou(t, g), calculate_area_iou(box, g.box)
                if ti >= tiou and ai >= aiou:
                    taken[j] = True
                    pairs.append((t, j, ti, ai))
                    break
    return pairs, [j for j, f in enumerate(taken) if not f]


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
