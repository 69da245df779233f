"""Shaders (interface of the reference's rendering/shader.py:12-263).

``Shader.forward(..., mode, degree)`` dispatches ``modes[mode % 8]`` exactly like the
reference; every mode is evaluated by the HIP kernel k_shade_fwd (rm_shade_forward) and, for
the three globally normalised modes, the second pass rm_shade_finish.  Inside RenderLoop the
same per-pixel code (shade_pixel in csrc/rm_kernels.h) runs fused at the end of the frame
kernel instead.
"""
from __future__ import annotations

import math
from pathlib import Path

import torch
import torch.nn as nn
from torch import Tensor

from .. import _abi
from ..ops import _f32c, _require_device

MODES = list(_abi.MODES)
_GLOBAL = (1, 2, 5)
_THREE_CHANNEL = (4, 6, 7)
_lib = _abi.lib


def default_cyclic_cmap(size: int = 4096) -> Tensor:
    """Procedural cyclic colormap with the shape, dtype and value range of the reference's
    data/cyclic_cmap.pt (float64 [4096,3], 0.237..1.0).  Used when that data file is not in
    the working directory."""
    ang = torch.arange(size, dtype=torch.float64) / size * math.tau
    phase = torch.tensor([0.0, math.tau / 3, 2 * math.tau / 3], dtype=torch.float64)
    return 0.6185 + 0.3815 * torch.cos(ang[:, None] - phase[None, :])


def load_cyclic_cmap() -> Tensor:
    """The reference loads ./data/cyclic_cmap.pt relative to the cwd (shader.py:177); do the
    same when it exists (weights_only: nothing in the file is executed)."""
    path = Path("./data/cyclic_cmap.pt")
    if path.is_file():
        return torch.load(path, weights_only=True)
    import warnings
    warnings.warn("ray_marching_amd: ./data/cyclic_cmap.pt (the reference's colormap, loaded relative to the working "
                  "directory at shader.py:177) is not here; the tangent / spin shaders will use a procedural cyclic "
                  "colormap with different colours.  Assign Shader.cyclic_cmap to use another table.", stacklevel=3)
    return default_cyclic_cmap()


def shade(mode: int, degree: int = 1, *, px_coords=None, orientation=None, frames=None, dirs=None,
          coords=None, normals=None, lap=None, dist=None, cmap=None, allreduce_minmax=None) -> Tensor:
    """Run one shader mode over tensors; returns [..., 1] (modes 0,1,2,3,5) or [..., 3]."""
    mode = mode % len(MODES)
    given = [t for t in (px_coords, dirs, coords, normals, lap, dist) if t is not None]
    if not given:
        raise ValueError("shade(): no per-pixel input")
    ref = given[0]
    _require_device(ref, "shader input")
    dev, in_dtype = ref.device, ref.dtype
    lead = ref.shape[:-1]
    n = 1
    for d in lead:
        n *= d
    per_cam = n // lead[0] if len(lead) > 1 else n

    def flat(t, ch):
        return None if t is None else _f32c(t.expand(*lead, ch)).reshape(-1, ch)

    bufs = dict(px=flat(px_coords, 3), dirs=flat(dirs, 3), coords=flat(coords, 3), normals=flat(normals, 3),
                lap=flat(lap, 1), dist=flat(dist, 1))
    q = None if orientation is None else _f32c(orientation).reshape(-1, 4)
    fr = None if frames is None else _f32c(frames).reshape(-1, 9)
    cm = None
    out_dtype = in_dtype if in_dtype in (torch.float32, torch.float16) else torch.float32
    if mode in (6, 7):
        if cmap is None:
            raise ValueError("tangent / spin shaders need a colormap")
        cm = cmap.detach()
        if cm.dtype not in (torch.float64, torch.float32, torch.float16):
            cm = cm.float()
        cm = cm.to(dev).contiguous()
        # brightness (input dtype) * colormap row: torch type promotion, e.g. fp32 * float64 -> float64 (shader.py:118)
        out_dtype = torch.promote_types(out_dtype, cm.dtype)
    first_dtype = torch.float32 if mode in _GLOBAL else out_dtype
    image = torch.empty((n, 3), dtype=first_dtype, device=dev)
    minmax = torch.empty(_abi.WORK_WORDS, dtype=torch.int32, device=dev) if mode in _GLOBAL else None
    with torch.cuda.device(dev):
        stream = _abi.current_stream(dev)
        if minmax is not None:
            _abi.check(_lib.rm_minmax_init(_abi.ptr(minmax), stream), "rm_minmax_init")
        _abi.check(_lib.rm_shade_forward(_abi.ptr(bufs["px"]), _abi.ptr(q), _abi.ptr(fr), _abi.ptr(bufs["dirs"]),
                                         _abi.ptr(bufs["coords"]), _abi.ptr(bufs["normals"]), _abi.ptr(bufs["lap"]),
                                         _abi.ptr(bufs["dist"]), _abi.ptr(image), _abi.dtype_code(first_dtype),
                                         _abi.ptr(minmax), _abi.ptr(cm), 0 if cm is None else cm.shape[0],
                                         0 if cm is None else _abi.dtype_code(cm.dtype), mode, degree, n, per_cam, stream),
                   "rm_shade_forward")
        if minmax is not None:
            if allreduce_minmax is not None:
                lohi = torch.empty(2, dtype=torch.float32, device=dev)
                _abi.check(_lib.rm_minmax_decode(_abi.ptr(minmax), _abi.ptr(lohi), stream), "rm_minmax_decode")
                allreduce_minmax(lohi)
                _abi.check(_lib.rm_minmax_encode(_abi.ptr(lohi), _abi.ptr(minmax), stream), "rm_minmax_encode")
            final = image if out_dtype == torch.float32 else torch.empty((n, 3), dtype=out_dtype, device=dev)
            _abi.check(_lib.rm_shade_finish(_abi.ptr(image), _abi.ptr(final), _abi.dtype_code(out_dtype), n,
                                            _abi.ptr(minmax), mode, _abi.DTYPE_F32, stream), "rm_shade_finish")
            image = final
    ch = 3 if mode in _THREE_CHANNEL else 1
    out = image.view(*lead, 3)[..., :ch]
    return out if mode in (6, 7) else out.to(in_dtype)


def _no_grad_inputs(*tensors):
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors):
        raise NotImplementedError("this shader has no backward kernel (the Lambertian, vignette and normal "
                                  "shaders do); differentiate through those or through RenderLoop modes 0 / 4")


class _PixelShade(torch.autograd.Function):
    """Differentiable wrapper for the per-pixel shaders with a VJP kernel (modes 0, 3, 4)."""

    @staticmethod
    def forward(ctx, dirs, normals, frames, mode: int):
        out = shade(mode, dirs=dirs, normals=normals, frames=frames)
        ctx.mode = mode
        ctx.save_for_backward(*(t for t in (dirs, normals, frames) if t is not None))
        ctx.have = (dirs is not None, normals is not None, frames is not None)
        return out

    @staticmethod
    def backward(ctx, grad):
        it = iter(ctx.saved_tensors)
        dirs, normals, frames = (next(it) if h else None for h in ctx.have)
        ref = dirs if dirs is not None else normals
        lead = torch.broadcast_shapes(*(t.shape[:-1] for t in (dirs, normals) if t is not None))
        dev = ref.device
        n = 1
        for d in lead:
            n *= d
        per_cam = n // lead[0] if len(lead) > 1 else n
        ch = 3 if ctx.mode == 4 else 1
        g = _f32c(grad.expand(*lead, ch)).reshape(-1, ch)
        d32 = None if dirs is None else _f32c(dirs.expand(*lead, 3)).reshape(-1, 3)
        n32 = None if normals is None else _f32c(normals.expand(*lead, 3)).reshape(-1, 3)
        f32 = None if frames is None else _f32c(frames).reshape(-1, 9)
        gd = torch.empty((n, 3), dtype=torch.float32, device=dev) if (dirs is not None and ctx.needs_input_grad[0]) else None
        gn = torch.empty((n, 3), dtype=torch.float32, device=dev) if (normals is not None and ctx.needs_input_grad[1]) else None
        with torch.cuda.device(dev):
            _abi.check(_lib.rm_shade_backward(_abi.ptr(d32), _abi.ptr(n32), _abi.ptr(f32), _abi.ptr(g), _abi.ptr(gd),
                                              _abi.ptr(gn), ctx.mode, n, per_cam, _abi.current_stream(dev)),
                       "rm_shade_backward")

        def fit(t, like):
            if t is None:
                return None
            t = t.view(*lead, 3).to(like.dtype)
            return t.sum_to_size(like.shape) if tuple(like.shape) != tuple(t.shape) else t

        return fit(gd, dirs) if dirs is not None else None, fit(gn, normals) if normals is not None else None, None, None


def _differentiable(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class LambertianShader(nn.Module):
    @torch.compiler.disable
    def forward(self, ray_directions: Tensor, surface_normals: Tensor) -> Tensor:
        if _differentiable(ray_directions, surface_normals):
            return _PixelShade.apply(ray_directions, surface_normals, None, 0)
        return shade(0, dirs=ray_directions, normals=surface_normals)


class DistanceShader(nn.Module):
    @torch.compiler.disable
    def forward(self, px_coords: Tensor, surface_coords: Tensor) -> Tensor:
        _no_grad_inputs(px_coords, surface_coords)
        return shade(1, px_coords=px_coords, coords=surface_coords)


class ProximityShader(nn.Module):
    @torch.compiler.disable
    def forward(self, surface_distances: Tensor) -> Tensor:
        _no_grad_inputs(surface_distances)
        return shade(2, dist=surface_distances)


class VignetteShader(nn.Module):
    @torch.compiler.disable
    def forward(self, ray_directions: Tensor, pixel_frames: Tensor) -> Tensor:
        if _differentiable(ray_directions):
            return _PixelShade.apply(ray_directions, None, pixel_frames, 3)
        return shade(3, dirs=ray_directions, frames=pixel_frames)


class NormalShader(nn.Module):
    @torch.compiler.disable
    def forward(self, surface_normals: Tensor) -> Tensor:
        if _differentiable(surface_normals):
            return _PixelShade.apply(None, surface_normals, None, 4)
        return shade(4, normals=surface_normals)


class LaplacianShader(nn.Module):
    @torch.compiler.disable
    def forward(self, surface_laplacian: Tensor) -> Tensor:
        _no_grad_inputs(surface_laplacian)
        return shade(5, lap=surface_laplacian)


def _orientation_from_conj(camera_orientation_conj: Tensor) -> Tensor:
    q = camera_orientation_conj.reshape(-1, 4)
    return torch.cat((q[:, :1], -q[:, 1:]), dim=-1)


class TangentShader(nn.Module):
    @torch.compiler.disable
    def forward(self, camera_orientation_conj: Tensor, ray_directions: Tensor, surface_normals: Tensor,
                cyclic_colourmap: Tensor, degree: int = 1) -> Tensor:
        _no_grad_inputs(ray_directions, surface_normals)
        return shade(6, degree, orientation=_orientation_from_conj(camera_orientation_conj), dirs=ray_directions,
                     normals=surface_normals, cmap=cyclic_colourmap)


class SpinShader(nn.Module):
    @torch.compiler.disable
    def forward(self, camera_orientation_conj: Tensor, surface_normals: Tensor, cyclic_colourmap: Tensor,
                degree: int = 1) -> Tensor:
        _no_grad_inputs(surface_normals)
        return shade(7, degree, orientation=_orientation_from_conj(camera_orientation_conj),
                     normals=surface_normals, cmap=cyclic_colourmap)


class Shader(nn.Module):
    """Dispatcher with the reference's call surface (shader.py:190-263)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("cyclic_cmap", load_cyclic_cmap())
        self.lambertian_shader = LambertianShader()
        self.normal_shader = NormalShader()
        self.tangent_shader = TangentShader()
        self.spin_shader = SpinShader()
        self.distance_shader = DistanceShader()
        self.proximity_shader = ProximityShader()
        self.vignette_shader = VignetteShader()
        self.laplacian_layer = LaplacianShader()

    @torch.compiler.disable
    def forward(self, px_coords: Tensor, camera_orientation: Tensor, pixel_frames: Tensor, ray_directions: Tensor,
                surface_coords: Tensor, surface_normals: Tensor, surface_laplacian: Tensor,
                surface_distances: Tensor, mode: int, degree: int) -> Tensor:
        if not isinstance(mode, int):
            raise NotImplementedError(f"{mode=} rendering mode not implemented.")
        m = mode % len(MODES)
        if m == 0:
            return self.lambertian_shader(ray_directions, surface_normals)
        if m == 3:
            return self.vignette_shader(ray_directions, pixel_frames)
        if m == 4:
            return self.normal_shader(surface_normals)
        _no_grad_inputs(px_coords, ray_directions, surface_coords, surface_normals, surface_laplacian, surface_distances)
        return shade(mode, degree, px_coords=px_coords, orientation=camera_orientation, frames=pixel_frames,
                     dirs=ray_directions, coords=surface_coords, normals=surface_normals, lap=surface_laplacian,
                     dist=surface_distances, cmap=self.cyclic_cmap)
