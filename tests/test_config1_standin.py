"""BASELINE config 1 (test/20220413Red.mp4 + config/m3det_normal.json) on the stand-in clip built from the ground truth the
reference holds for its missing video (tests/config1_chain.py, metdetpy_amd.synth.Config1Clip).

CPU: the oracle chain finds the annotated meteor under evaluate.py's own rule (score > 0.5, tIoU >= 0.3, IoU >= 0.3).
GPU (-m gpu): the product chain (pipeline.detect_clip: frame sums, preprocessing, detector, native collector, all through the
C-ABI) gives the oracle chain's exposure decision, thresholds, lit counts, segments and records -- field for field -- and matches
the annotation too.
"""
import numpy as np
import pytest

from metdetpy_amd.metstruct import MDRF, BasicInfo
from metdetpy_amd.pipeline import match_annotation, parse_resize_param, to_mdrf
from metdetpy_amd.utils import ts2frame
from tests import config1_chain as C1


def test_parse_resize_param_is_the_loaders():
    # utils.py:562-617 (hand-computed from the reference's rule: one integer = the long side, the other follows the aspect ratio)
    assert parse_resize_param(960, [3840, 2160]) == [960, 540]
    assert parse_resize_param(960, [2160, 3840]) == [540, 960]
    assert parse_resize_param("960x540", [1920, 1080]) == [960, 540]
    assert parse_resize_param("480", [1920, 1080]) == [480, 270]
    assert parse_resize_param([-1, 360], [1920, 1080]) == [640, 360]
    assert parse_resize_param(None, [1920, 1080]) == [1920, 1080]
    assert parse_resize_param([0, 0], [1920, 1080]) == [1920, 1080]
    with pytest.raises(TypeError):
        parse_resize_param(1.5, [1920, 1080])


def _as_mdrf(records, rt, total, fps):
    info = BasicInfo("ThreadVideoLoader", "config1_standin", "mask-east.jpg", 0, int(total / fps * 1000), list(rt.raw_size),
                     list(rt.runtime_size), rt.exp_time, total, fps)
    return to_mdrf(records, info)


def test_oracle_chain_finds_the_annotated_meteor():
    """The reference-held ground truth is the known answer: one meteor, 2.4 s -> 4.4 s, [433, 147] -> [374, 222] at 1168x655."""
    oc = C1.oracle_chain()
    t = C1.truth()
    rt = oc["rt"]
    assert (rt.exp_frame, rt.eq_int_fps, oc["n"]) == (2, 12, 12) and rt.runtime_size == [960, 540]
    doc = _as_mdrf(oc["records"], rt, 150, 25.0)
    pairs, missed = match_annotation(doc, t["annotation"], 25.0, [3840, 2160])
    assert not missed and len(pairs) == 1, (pairs, missed)
    tgt, j, tiou, aiou = pairs[0]
    assert tgt.category == "METEOR" and tgt.score > 0.5 and tiou >= 0.3 and aiou >= 0.3
    # nothing else positive in the clip: one annotated meteor, one positive prediction
    positives = [x for r in oc["records"] for x in r.target if x.score > 0.5 and x.category in rt.positive_category_list]
    assert len(positives) == 1
    # the reference's second record of the same meteor (test/clip_test.json: 2.56 s -> 4.32 s, [1412, 496] -> [1232, 728] at the
    # video's own 3840x2160): the detection starts within two exposures of it and its end points lie within 16 source pixels
    c = t["clip_test_entry_0"]
    assert abs(tgt.start_frame - ts2frame(c["time"][0], 25.0)) <= 2 * rt.exp_frame
    for got, want in ((tgt.pt1, c["target"][0]["pt1"]), (tgt.pt2, c["target"][0]["pt2"])):
        assert max(abs(got[0] - want[0]), abs(got[1] - want[1])) <= 16, (got, want)


@pytest.mark.gpu
def test_gpu_chain_equals_the_oracle_chain_and_finds_the_meteor():
    from metdetpy_amd import exposure as EX
    from metdetpy_amd.imgproc import Preprocessor
    from metdetpy_amd.pipeline import detect_clip
    oc = C1.oracle_chain()
    rd = C1.ClipReader()
    cfg, mask = oc["cfg"], oc["mask"]
    # the pieces: frame sums (the exposure decision hangs on them) and the loader chain on two frames
    assert np.array_equal(EX.frame_sums(rd(0, 12)), oc["sums"][:12])
    pre = Preprocessor(rd.hw, (960, 540), mask)
    assert np.array_equal(pre(rd(0, 1)[0]), oc["gray_first"]) and np.array_equal(pre(rd(80, 81)[0]), oc["gray_mid"])
    pre.close()
    doc, rt, info = detect_clip(rd, rd.total, rd.fps, rd.hw, cfg, mask, batch=32, video="config1_standin", mask_name="mask-east.jpg")
    assert rt == oc["rt"], (rt, oc["rt"])
    assert len(info) == len(oc["rows"]) == 75
    for i, (thr, lit, n_raw, _lines) in enumerate(oc["rows"]):
        assert (int(info[i]["bi_threshold"]), int(info[i]["lit_count"]), int(info[i]["n_raw"])) == (thr, lit, n_raw), i
    assert [r.to_dict() for r in doc.results] == [r.to_dict() for r in oc["records"]]
    assert doc.anno_size == [3840, 2160] and doc.type == "prediction" and doc.config is cfg
    pairs, missed = match_annotation(doc, C1.truth()["annotation"], rd.fps, [3840, 2160])
    assert not missed and len(pairs) == 1 and pairs[0][0].category == "METEOR"
    assert isinstance(MDRF.from_dict(doc.to_dict()), MDRF)  # the document round-trips through the reference's schema
