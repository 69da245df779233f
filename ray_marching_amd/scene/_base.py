"""Common base of the SDF nodes: forward() = compile the subtree, run the HIP evaluator."""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from ..compiler import compiled_for
from ..ops import SDFEval, live_params


class SDFNode(nn.Module):
    """An SDF scene-graph node.  ``node(query[..., 3]) -> [..., 1]`` like the reference's
    modules (scene/primitives.py, scene/transformations.py), differentiable w.r.t. the
    query points and every nn.Parameter below the node, evaluated by rm_sdf_forward /
    rm_sdf_backward."""

    _rm_kind = None

    def forward(self, query_positions: Tensor) -> Tensor:
        return self._evaluate(query_positions)

    @torch.compiler.disable      # under torch.compile (main.py:44) the launch stays the eager ctypes call
    def _evaluate(self, points: Tensor) -> Tensor:
        if points.shape[-1] != 3:
            raise ValueError(f"query points must be [..., 3], got {tuple(points.shape)}")
        cs = compiled_for(self)
        return SDFEval.apply(live_params(cs, points.device, points), points, cs)


class SDFCoordsNode(SDFNode):
    """The four combinators whose ``forward`` names its argument ``query_coords`` in the reference
    (scene/transformations.py:67, 90, 117, 131), so ``sdf(query_coords=p)`` works here as there."""

    def forward(self, query_coords: Tensor) -> Tensor:
        return self._evaluate(query_coords)
