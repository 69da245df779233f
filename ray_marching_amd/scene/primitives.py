"""SDF leaf nodes (interface of the reference's scene/primitives.py:6-102).

Same class names, constructor arguments and nn.Parameter names; the distance
formulas live in csrc/rm_device.h (fwd_op / bwd_op), not here.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ._base import SDFNode


def _param(value):
    return nn.Parameter(torch.tensor(value, dtype=torch.float32))


class SDFSphere(SDFNode):
    """|p| - radius."""
    _rm_kind = "sphere"

    def __init__(self, radius: float):
        super().__init__()
        self.radius = _param(radius)


class SDFBox(SDFNode):
    """Axis-aligned box with the given half side lengths."""
    _rm_kind = "box"

    def __init__(self, halfsides: tuple[float, float, float]):
        super().__init__()
        self.halfsides = _param(halfsides)


class SDFPlane(SDFNode):
    """Half space x < 0 (distance = p.x); no parameters."""
    _rm_kind = "plane"

    def __init__(self):
        super().__init__()


class SDFLine(SDFNode):
    """Capsule from ``start`` to ``end`` with ``radius``."""
    _rm_kind = "line"

    def __init__(self, start, end, radius: float):
        super().__init__()
        self.start = _param(start)
        self.end = _param(end)
        self.radius = _param(radius)


class SDFDisk(SDFNode):
    """Flat disk: axis x, radius measured in the yz plane."""
    _rm_kind = "disk"

    def __init__(self, radius: float):
        super().__init__()
        self.radius = _param(radius)


class SDFTorus(SDFNode):
    """Torus with ring radius ``radius1`` in the xz plane and tube radius ``radius2``."""
    _rm_kind = "torus"

    def __init__(self, radius1: float, radius2: float) -> None:
        super().__init__()
        self.radius1 = _param(radius1)
        self.radius2 = _param(radius2)
