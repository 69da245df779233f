"""Quaternion helpers with the call surface of the reference's quaternion.py:6-124.

On the hot path these operations are fused into the HIP kernels (csrc/rm_device.h: cross,
qrot, k_camera_fwd).  The functions here serve pose bookkeeping on tiny [N,4] / [N,3]
tensors (camera state updates), which is host-side plumbing, not the data-parallel path.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor


def cross_product(U: Tensor, V: Tensor) -> Tensor:
    a, b, c = U.unbind(-1)
    d, e, f = V.unbind(-1)
    return torch.stack((b * f - c * e, c * d - a * f, a * e - b * d), dim=-1)


def multiply(p: Tensor, q: Tensor) -> Tensor:
    """Hamilton product p (x) q, with the operand and summation order of the reference (quaternion.py:38-46):
    each component is a three-term sum (rounded left to right) minus or plus one product."""
    p0, p1, p2, p3 = p.unbind(-1)
    q0, q1, q2, q3 = q.unbind(-1)

    def s3(a, b, c):
        return torch.stack((a, b, c), dim=-1).sum(-1)

    return torch.stack((p0 * q0 - s3(p1 * q1, p2 * q2, p3 * q3),
                        s3(p0 * q1, p1 * q0, p2 * q3) - p3 * q2,
                        s3(p0 * q2, p2 * q0, p3 * q1) - p1 * q3,
                        s3(p0 * q3, p1 * q2, p3 * q0) - p2 * q1), dim=-1)


def conjugate(q: Tensor) -> Tensor:
    return torch.cat((q[..., :1], -q[..., 1:]), dim=-1)


def rotation(V: Tensor, q: Tensor) -> Tensor:
    """V + w t + qv x t with t = 2 qv x V (q is used as given, not normalised)."""
    qv = q[..., 1:]
    t = 2 * cross_product(qv, V)
    return (cross_product(qv, t) + q[..., :1] * t) + V


def to_versor(V: Tensor) -> Tensor:
    """(sqrt(1 - |V|^2), V)."""
    w = (1 - V.pow(2).sum(dim=-1, keepdim=True)).sqrt()
    return torch.cat((w, V), dim=-1)


class QuaternionToSO3(nn.Module):
    """Quaternion [...,4] -> rotation matrices [-1,3,3] (row major)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("pairs", torch.tensor([(i, j) for i in range(4) for j in range(i, 4)]))

    def forward(self, q: Tensor) -> Tensor:
        w, x, y, z = q.unbind(-1)
        rows = (w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (w * y + x * z),
                2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                2 * (x * z - w * y), 2 * (w * x + y * z), w * w - x * x - y * y + z * z)
        return torch.stack(rows, dim=-1).view(-1, 3, 3)
