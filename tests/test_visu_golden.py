"""M3Detector.visu() against the records the REFERENCE's own visu() returned (tests/golden/g11_visu.json: class names, field names,
defaults, sync lists, texts, the overlay image and the std-roi rectangle)."""
import dataclasses
import json
import os
import types

import numpy as np
import pytest

from metdetpy_amd.detector import visu_records


def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "g11_visu.json")))


@pytest.mark.parametrize("ci", range(3))
def test_visu_records_equal_reference(golden_dir, ci):
    c = _cases(golden_dir)[ci]
    st = c["state"]
    stack = types.SimpleNamespace(snr=st["snr"], std_roi=tuple(c["std_roi"]) if c["std_roi"] else None)
    items = visu_records(np.array(st["dst"], np.uint8), stack, st["bi_threshold"], st["bi_threshold_float"], st["lines_num"],
                         st["filtered_line_num"], st["dst_sum"])
    assert len(items) == len(c["visu"])
    for got, want in zip(items, c["visu"]):
        assert type(got).__name__ == want["class"]
        d = dataclasses.asdict(got)
        if d.get("img") is not None:
            d["img"] = np.asarray(d["img"]).tolist()
        d = json.loads(json.dumps(d))  # tuples -> lists, as the fixture went through JSON
        assert list(d.keys()) == list(want["fields"].keys()), want["class"]
        assert d == want["fields"], (want["class"], want["fields"]["name"])


@pytest.mark.gpu
def test_visu_on_a_live_detector():
    from metdetpy_amd import M3Detector
    from metdetpy_amd.synth import SkyStream
    from tests.helpers import binary_cfg
    H, W = 135, 240
    det = M3Detector(1, 8, np.ones((H, W), np.uint8), 10, binary_cfg(), None)
    s = SkyStream(H, W, 5, speed=6, streak_period=10)
    for _ in range(20):
        det.update(s.next())
        det.detect()
    items = det.visu()
    assert [i.name for i in items] == ["mix_bg", "std_value", "bi_value", "lines_num", "area_ratio", "lines_warning", "std_roi_area"]
    assert items[0].img.shape == (H, W) and set(np.unique(items[0].img)) <= {0, 1}
    assert items[1].text_list[0].text == f"STD:{det.stack.snr:.4f}"
    y1x1, y2x2 = items[-1].pair_list[0].dot_pair
    assert [y1x1[1], y1x1[0], y2x2[1], y2x2[0]] == list(det.stack.std_roi)
    det.close()
