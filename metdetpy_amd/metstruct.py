"""Config / result records of the hot path's boundary (mirrors the field names of the reference's
metstruct.py:61-70, 142-246, 357-425 so that configs and MDRF records are interchangeable).

`from_dict` is a small recursive parser for nested dataclasses (the reference uses dacite for this).
"""
from __future__ import annotations

import dataclasses
import json
import typing
from typing import Any, Optional, Union

import numpy as np


def _build(tp, value):
    origin = typing.get_origin(tp)
    if dataclasses.is_dataclass(tp) and isinstance(value, dict):
        hints = typing.get_type_hints(tp)
        kwargs = {}
        for f in dataclasses.fields(tp):
            if f.name in value:
                kwargs[f.name] = _build(hints[f.name], value[f.name])
        return tp(**kwargs)
    if origin is Union:
        for arg in typing.get_args(tp):
            if arg is type(None):
                if value is None:
                    return None
                continue
            try:
                return _build(arg, value)
            except (TypeError, ValueError):
                continue
        return value
    if origin in (list, typing.List) and isinstance(value, list):
        (inner, ) = typing.get_args(tp) or (Any, )
        return [_build(inner, v) for v in value]
    return value


@dataclasses.dataclass
class DictAble:
    """to_dict / to_json / from_dict helpers (reference: metstruct.py:76-124)."""

    def _plain(self, name: str, full: bool):
        v = getattr(self, name)
        if isinstance(v, DictAble):
            return v.to_dict(full)
        if isinstance(v, (list, tuple)):
            return [x.to_dict(full) if isinstance(x, DictAble) else x for x in v]
        if isinstance(v, np.floating):
            return float(v)
        if isinstance(v, np.integer):
            return int(v)
        return v

    def to_dict(self, full: bool = True) -> dict:
        hidden = [] if full else list(getattr(self, "exclude_attrs", []))
        return {f.name: self._plain(f.name, full) for f in dataclasses.fields(self)
                if f.name != "exclude_attrs" and f.name not in hidden}

    def to_json(self, full: bool = True) -> str:
        return json.dumps(self.to_dict(full))

    @classmethod
    def from_dict(cls, data: dict):
        return _build(cls, data)

    @classmethod
    def from_json_file(cls, path: str):
        with open(path, "r", encoding="utf-8") as f:
            return cls.from_dict(json.load(f))


@dataclasses.dataclass
class Box:
    x1: int
    y1: int
    x2: int
    y2: int

    @classmethod
    def from_list(cls, c):
        assert len(c) == 4
        x1, y1, x2, y2 = c
        return cls(max(0, min(x1, x2)), max(0, min(y1, y2)), max(x1, x2), max(y1, y2))

    @classmethod
    def from_pts(cls, p1, p2):
        return cls.from_list([*p1, *p2])


@dataclasses.dataclass
class RuntimeParams:
    fps: float
    exp_frame: int
    eq_fps: float
    eq_int_fps: int
    exp_time: float
    runtime_size: list
    raw_size: list
    positive_category_list: list


@dataclasses.dataclass
class BinaryCoreCfg(DictAble):
    adaptive_bi_thre: bool
    init_value: int
    sensitivity: str
    area: float
    interval: int


@dataclasses.dataclass
class HoughLineCfg(DictAble):
    threshold: int
    min_len: int
    max_gap: int


@dataclasses.dataclass
class DynamicCfg(DictAble):
    dy_mask: bool
    window_sec: float


@dataclasses.dataclass
class BinaryCfg(DictAble):
    binary: BinaryCoreCfg
    hough_line: HoughLineCfg
    dynamic: DynamicCfg


@dataclasses.dataclass
class DetectorCfg(DictAble):
    name: str
    window_sec: float
    cfg: BinaryCfg


@dataclasses.dataclass
class MeteorCfg(DictAble):
    min_len: float
    max_interval: float
    time_range: list
    speed_range: list
    drct_range: list
    det_thre: float
    thre2: int


@dataclasses.dataclass
class ModelCfg(DictAble):
    """recheck_cfg.model as the presets carry it (metstruct.py:342-353).  Carried through to the MDRF document only: the YOLO
    re-check itself is outside this path."""
    name: str
    weight_path: str
    dtype: str
    nms: bool
    warmup: bool
    pos_thre: float
    nms_thre: float
    multiscale_pred: int
    multiscale_partition: int
    providers_key: Optional[str] = "default"


@dataclasses.dataclass
class RecheckSwitch(DictAble):
    """recheck_cfg (metstruct.py:408-411).  Only the switch is honoured (and must be off): the YOLO recheck is outside this
    path; `model` is kept so that a preset survives a round trip into the MDRF document."""
    switch: bool = False
    model: Optional[ModelCfg] = None


RecheckCfg = RecheckSwitch


@dataclasses.dataclass
class CollectorCfg(DictAble):
    meteor_cfg: MeteorCfg
    recheck_cfg: RecheckSwitch = dataclasses.field(default_factory=RecheckSwitch)
    positive_cfg: dict = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class LoaderCfg(DictAble):
    """metstruct.py:329-339 (the loader itself is outside this path; the block travels into the MDRF document)."""
    name: str
    wrapper: str
    resize: Union[list, int, str]
    exp_time: Union[float, str]
    merge_func: str
    grayscale: bool
    upper_bound: Optional[float] = None
    continue_on_err: bool = False
    hwaccel: Optional[str] = None


@dataclasses.dataclass
class MainDetectCfg(DictAble):
    """A whole preset file, config/m3det_*.json (metstruct.py:421-425); keys the schema does not declare are dropped."""
    loader: LoaderCfg
    detector: DetectorCfg
    collector: CollectorCfg


@dataclasses.dataclass
class BasicInfo(DictAble):
    """metstruct.py:127-139."""
    loader: str
    video: str
    mask: Optional[str]
    start_time: int
    end_time: int
    resolution: list
    runtime_resolution: list
    exp_time: float
    total_frames: int
    fps: float
    desc: Optional[str] = None


@dataclasses.dataclass
class MDTarget(DictAble):
    start_frame: int
    start_time: str
    end_time: str
    last_activate_frame: int
    last_activate_time: str
    duration: int
    speed: float
    dist: float
    fix_dist: float
    fix_speed: float
    fix_motion_duration: float
    fix_duration: float
    num_pts: int
    category: str
    pt1: list
    pt2: list
    drct_loss: float
    score: float
    real_dist: float
    center_point_list: list = dataclasses.field(default_factory=list)
    raw_score: Optional[float] = None
    recheck_score: Optional[float] = None
    relative_brightness: Optional[float] = None
    aesthetic_score: Optional[float] = None
    exclude_attrs: list = dataclasses.field(default_factory=lambda: ["center_point_list"])


@dataclasses.dataclass
class SingleMDRecord(DictAble):
    start_time: str
    end_time: str
    video_size: list
    target: list
    start_frame: Optional[int] = None
    end_frame: Optional[int] = None

    @classmethod
    def from_target(cls, target: MDTarget, video_size):
        return cls.from_target_list([target], video_size)

    @classmethod
    def from_target_list(cls, targets, video_size):
        return cls(start_frame=min(t.start_frame for t in targets), start_time=min(t.start_time for t in targets),
                   end_time=max(t.last_activate_time for t in targets),
                   end_frame=max(t.last_activate_frame for t in targets), video_size=video_size, target=targets)


@dataclasses.dataclass
class MDRF(DictAble):
    """The Meteor Detection Recording Format document `detect_video` returns and `MetDetPy.py` writes with
    `json.dump(result.to_dict(), ...)` (metstruct.py:428-441, MetDetPy.py:245-250, 395-398)."""
    version: str
    basic_info: BasicInfo
    config: Optional[MainDetectCfg]
    type: str
    anno_size: Optional[list]
    results: list
    performance: Optional[dict] = None

