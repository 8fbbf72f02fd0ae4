"""End-to-end on a synthetic sky: GPU detector + collector vs CPU oracle detector + the same collector ->
records identical field for field, and the injected streaks are found under the reference's own matching rule
(evaluate.py: time-IoU >= 0.3 and box-IoU >= 0.3)."""
import numpy as np
import pytest

from oracle.collector_oracle import MeteorCollector
from metdetpy_amd.metstruct import Box, CollectorCfg, MDTarget, MeteorCfg, RecheckSwitch, RuntimeParams
from metdetpy_amd.metstruct import BasicInfo
from metdetpy_amd.pipeline import calculate_area_iou, calculate_time_iou, detect_stream, to_mdrf
from metdetpy_amd.synth import SkyStream
from metdetpy_amd.utils import get_num_class
from oracle import c_oracle as co
from oracle import m3_oracle as mo
from tests.helpers import binary_cfg, oracle_kwargs

pytestmark = pytest.mark.gpu


def test_stream_records_equal_oracle_pipeline_and_find_the_streaks():
    H, W, n, fps = 270, 480, 10, 10.0
    total = 260
    stream = SkyStream(H, W, 4242, speed=3.0, streak_period=90, streak_life=14)
    frames, truth = [], []   # truth: (first frame, last frame, box) of each injected streak
    for t in range(total):
        was = stream._streak
        frames.append(stream.next())
        if was is None and stream._streak is not None:      # a streak was born in this frame
            p0, v, _ = stream._streak
            a, b = p0, p0 + v * 14
            truth.append((t, t + 13, Box.from_pts([int(a[0]), int(a[1])], [int(b[0]), int(b[1])])))
    assert len(truth) == 3
    mask = np.ones((H, W), np.uint8)
    cfg = binary_cfg()
    ccfg = CollectorCfg(MeteorCfg(8, 2, [0, 8], [0.5, 40], [0, 0.6], 0.5, 2048), RecheckSwitch(False), {})
    rt = RuntimeParams(fps=fps, exp_frame=1, eq_fps=fps, eq_int_fps=10, exp_time=0.1, runtime_size=[W, H], raw_size=[W * 2, H * 2],
                       positive_category_list=["METEOR"])
    records, info = detect_stream(frames, mask, cfg, ccfg, rt, batch=37)
    # the same loop with the CPU oracle as the detector
    orc = mo.M3Oracle(n, mask, hough_fn=co.hough_lines_p, **oracle_kwargs(cfg))
    col = MeteorCollector(ccfg, rt, None, None)
    for it, f in enumerate(frames):
        orc.update(f)
        lines, cates = orc.detect()
        assert int(info[it]["bi_threshold"]) == orc.bi_threshold and int(info[it]["n_raw"]) == orc.lines_num
        if len(lines) or it % rt.eq_int_fps == 0:
            col.update(it, lines=lines, cates=cates)
    col.clear()
    want = [r.to_dict() for r in col.met_exporter.meteor_list]
    got = [r.to_dict() for r in records]
    assert got == want
    info_blk = BasicInfo("synthetic", "sky", None, 0, int(len(frames) / fps * 1000), [W * 2, H * 2], [W, H], rt.exp_time, len(frames), fps)
    doc = to_mdrf(records, info_blk).to_dict()
    assert doc["results"] == want and doc["anno_size"] == [W * 2, H * 2] and doc["type"] == "prediction"
    # the oracle-side Python collector driven frame by frame through the same loop gives the same records
    records_n, info_n = detect_stream(frames, mask, cfg, ccfg, rt, batch=64, collector_cls=MeteorCollector)
    assert [r.to_dict() for r in records_n] == want and np.array_equal(info_n["n_raw"], info["n_raw"])
    # every injected streak is matched by a positive record (evaluate.py rule)
    targets = [t for r in records for t in r.target if t.category == "METEOR" and t.score > 0.5]
    assert targets, [t.category for r in records for t in r.target]
    scale = 2
    for (f0, f1, box) in truth:
        gt = MDTarget(start_frame=f0, start_time="", end_time="", last_activate_frame=f1 + n, last_activate_time="", duration=0,
                      speed=0, dist=0, fix_dist=0, fix_speed=0, fix_motion_duration=0, fix_duration=0, num_pts=0,
                      category="METEOR", pt1=[box.x1 * scale, box.y1 * scale], pt2=[box.x2 * scale, box.y2 * scale],
                      drct_loss=0, score=1, real_dist=0)
        hit = [t for t in targets if calculate_time_iou(t, gt) >= 0.3 and
               calculate_area_iou(Box.from_pts(t.pt1, t.pt2), Box.from_pts(gt.pt1, gt.pt2)) >= 0.3]
        assert hit, (f0, f1, box, [(t.start_frame, t.last_activate_frame, t.pt1, t.pt2) for t in targets])
