"""Camera, marcher and normals modules (interface of the reference's
rendering/ray_marching.py:9-125) backed by the HIP entry points
rm_camera_forward, rm_march_forward/backward and rm_normals_forward/backward."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import ops
from ..compiler import compiled_for
from ..quaternion import QuaternionToSO3


def pinhole_grid(num_cameras: int, px_width: int, px_height: int, focal_length: float,
                 sensor_width: float, sensor_height: float):
    """Camera-frame ray origins on the sensor plane z=0 and unit directions through the
    focus (0,0,-f).  Pixel centres come from ``F.affine_grid(align_corners=False)`` so the
    fp32 values are the ones the reference's constructor stores (ray_marching.py:26-50)."""
    theta = torch.zeros(num_cameras, 2, 3, dtype=torch.float32)
    theta[:, 0, 0] = sensor_width / 2
    theta[:, 1, 1] = -sensor_height / 2
    xy = F.affine_grid(theta, (num_cameras, 1, px_height, px_width), align_corners=False)
    origins = torch.cat([xy, torch.zeros_like(xy[..., :1])], dim=-1)
    focus = torch.tensor([0.0, 0.0, -focal_length], dtype=torch.float32)
    directions = F.normalize(origins - focus, p=2, dim=-1, eps=0)
    return origins, directions, theta, focus


class PinholeCamera(nn.Module):
    """forward(orientation[N,4], translation[N,3]) -> (ray_pos, R[N,3,3], ray_pos, ray_dirs)."""

    def __init__(self, num_cameras: int, px_width: int, px_height: int, focal_length: float,
                 sensor_width: float, sensor_height: float, rows=None):
        """``rows=(r0, r1)`` (extension for row-tiled multi-GPU rendering): keep only that band of the ray
        buffers -- the values are the full frame's rows r0..r1, bit for bit -- so a rank that renders one band
        of an 8K frame holds 100 MB of camera buffers instead of 800 MB."""
        super().__init__()
        self.num_cameras = num_cameras
        self.focal_length = focal_length
        self.sensor_width = sensor_width
        self.sensor_height = sensor_height
        self.size = (num_cameras, 1, px_height, px_width)
        origins, directions, theta, focus = pinhole_grid(num_cameras, px_width, px_height, focal_length,
                                                         sensor_width, sensor_height)
        self.rows = (0, px_height) if rows is None else (int(rows[0]), int(rows[1]))
        if not (0 <= self.rows[0] < self.rows[1] <= px_height):
            raise ValueError(f"rows {rows} outside the frame of {px_height} rows")
        if rows is not None:
            origins = origins[:, self.rows[0]:self.rows[1]].contiguous()
            directions = directions[:, self.rows[0]:self.rows[1]].contiguous()
        # buffer names follow the reference so state_dicts interchange.  The reference registers `focus`, `theta` and
        # `pixel_frames` as expanded (overlapping) views, which load_state_dict cannot write into once
        # num_cameras > 1; here they own their memory, so a reference checkpoint loads for any N.
        self.register_buffer("focus", focus.view(1, 1, 1, 3).expand(num_cameras, 1, 1, 3).clone())
        self.register_buffer("theta", theta)
        self.register_buffer("ray_positions", origins)
        self.register_buffer("ray_directions", directions)
        self.register_buffer("pixel_frames", torch.eye(3)[None, None, None, :2, :].expand(num_cameras, 1, 1, 2, 3).clone())
        self.quaternion_to_so3 = QuaternionToSO3()

    @torch.compiler.disable
    def forward(self, orientation: Tensor, translation: Tensor):
        pos, frames, dirs = ops.camera_forward(self.ray_positions, self.ray_directions, orientation, translation)
        return (pos, frames, pos, dirs)


class SDFMarcher(nn.Module):
    """p <- scene(p) * v + p, ``marching_steps`` times (reference ray_marching.py:72-84).

    ``early_out`` lets a wave stop once all its rays sit on a bit-exact fixed point or
    2-cycle of the march map; the returned positions are unchanged by it."""

    def __init__(self, sdf_scene: nn.Module, early_out: bool = True):
        super().__init__()
        self.sdf_scene = sdf_scene
        self.early_out = early_out

    @torch.compiler.disable
    def forward(self, ray_positions: Tensor, ray_directions: Tensor, marching_steps: int = 32) -> Tensor:
        cs = compiled_for(self.sdf_scene)
        return ops.March.apply(ops.live_params(cs, ray_positions.device, ray_positions, ray_directions), ray_positions, ray_directions, cs,
                               int(marching_steps), ops.default_flags(self.early_out))


def tetrahedron_constants(normals_eps: float):
    """Tap offsets [4,3] (regular tetrahedron, edge-normalised, scaled by eps) and the inverse of
    the 3x3 matrix of offsets relative to tap 0 (reference ray_marching.py:96-113)."""
    a = 0.5 ** 0.5
    taps = torch.tensor([[1.0, 0.0, -a], [-1.0, 0.0, -a], [0.0, 1.0, a], [0.0, -1.0, a]])
    taps = F.normalize(taps, dim=-1, p=2, eps=0.0) * normals_eps
    rel = taps[1:] - taps[:1]
    return taps, rel, rel.inverse()


class SDFNormals(nn.Module):
    """forward(coords[...,3]) -> (normals[...,3], laplacian[...,1]): 4-tap tetrahedral finite
    difference solved by a precomputed 3x3 inverse, Laplacian from the centre tap."""

    def __init__(self, sdf_scene: nn.Module, normals_eps: float = 1e-3):
        super().__init__()
        self.sdf_scene = sdf_scene
        self.normals_eps = normals_eps
        taps, rel, inv = tetrahedron_constants(normals_eps)
        self.register_buffer("offsets", taps)
        self.register_buffer("relative_offsets", rel)
        self.register_buffer("offsets_inverse", inv)
        self._tetra = None

    def tetra(self):
        key = (self.offsets.dtype, self.offsets._version)
        if self._tetra is None or self._tetra[0] != key:
            self._tetra = (key, ops.make_tetra(self.offsets, self.offsets_inverse, self.normals_eps))
        return self._tetra[1]

    @torch.compiler.disable
    def forward(self, surface_coords: Tensor):
        cs = compiled_for(self.sdf_scene)
        return ops.Normals.apply(ops.live_params(cs, surface_coords.device, surface_coords), surface_coords, cs, self.tetra())
