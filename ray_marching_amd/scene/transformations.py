"""SDF combinators (interface of the reference's scene/transformations.py:8-132)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ._base import SDFCoordsNode, SDFNode


def _param(value):
    return nn.Parameter(torch.tensor(value, dtype=torch.float32))


class SDFAffineTransformation(SDFNode):
    """child(rot(p - translation, conj(orientation))); the quaternion is used as given
    (not normalised), exactly like the reference."""
    _rm_kind = "affine"

    def __init__(self, sdf: nn.Module, orientation, translation):
        super().__init__()
        self.sdf = sdf
        self.translation = _param(translation)
        self.orientation = _param(orientation)


class SDFSmoothUnion(SDFCoordsNode):
    """-logsumexp(-k d_i) / k over the children."""
    _rm_kind = "smooth_union"

    def __init__(self, sdfs, blend_k: float):
        super().__init__()
        self.sdfs = nn.ModuleList(sdfs)
        self.blend_k = _param(blend_k)


class SDFUnion(SDFCoordsNode):
    """min over the children."""
    _rm_kind = "union"

    def __init__(self, sdfs):
        super().__init__()
        self.sdfs = nn.ModuleList(sdfs)


class SDFRounding(SDFCoordsNode):
    """d - rounding."""
    _rm_kind = "rounding"

    def __init__(self, sdf: nn.Module, rounding: float):
        super().__init__()
        self.sdf = sdf
        self.rounding = _param(rounding)


class SDFOnion(SDFCoordsNode):
    """|d| - radius (a shell of the child)."""
    _rm_kind = "onion"

    def __init__(self, sdf: nn.Module, radius: float):
        super().__init__()
        self.sdf = sdf
        self.radius = _param(radius)
