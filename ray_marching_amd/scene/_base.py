"""Common base of the SDF nodes: forward() = compile the subtree, run the HIP evaluator."""
from __future__ import annotations

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
        if query_positions.shape[-1] != 3:
            raise ValueError(f"query_positions must be [..., 3], got {tuple(query_positions.shape)}")
        cs = compiled_for(self)
        return SDFEval.apply(live_params(cs, query_positions.device, query_positions), query_positions, cs)
