"""The attribute records `detector.visu()` hands to the reference's visual manager (MetLib/metvisu.py:74-260): same class names,
field names, defaults and sync lists, so that `MetVisu.display_a_frame` can take them unchanged.  Rendering itself (cv2 drawing,
windows) is outside this path (out of scope): these are plain records."""
from __future__ import annotations

import dataclasses
from typing import Any, Optional


@dataclasses.dataclass
class BaseVisuAttrs:
    name: str
    sync_attributes: list = dataclasses.field(default_factory=list)


@dataclasses.dataclass
class ImgVisuAttrs(BaseVisuAttrs):
    weight: Optional[float] = None
    img: Any = None
    color: Any = None
    sync_attributes: list = dataclasses.field(default_factory=lambda: ["weight", "img", "color"])


@dataclasses.dataclass
class DrawVisuAttrs(BaseVisuAttrs):
    color: Any = None
    thickness: Optional[int] = None
    sync_attributes: list = dataclasses.field(default_factory=lambda: ["color", "thickness"])


@dataclasses.dataclass
class SquareColorPair:
    dot_pair: Any = None
    color: Any = None
    thickness: Optional[int] = None


@dataclasses.dataclass
class DrawRectVisu(DrawVisuAttrs):
    pair_list: Optional[list] = None


@dataclasses.dataclass
class TextColorPair:
    text: Optional[str] = None
    position: Any = None
    color: Any = None


@dataclasses.dataclass
class TextVisu(BaseVisuAttrs):
    text_list: Optional[list] = None
    position: Any = None
    color: Any = None
    font_face: Optional[int] = None
    font_scale: Optional[float] = None
    font_thickness: Optional[int] = None
    position_flag: bool = False
    sync_attributes: list = dataclasses.field(default_factory=lambda: ["position", "color", "font_face", "font_scale", "font_thickness"])
