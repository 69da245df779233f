"""RenderLoop with the call surface of the reference's control.py:197-258.

The reference chains camera -> marcher -> scene -> normals -> shader as separate module calls
(hundreds of ATen launches per frame).  Here ``forward`` issues ONE fused HIP kernel
(rm_render_forward: k_render_fwd) plus, for shader modes 1/2/5, the tiny normalisation pass.
The sub-modules are still constructed (same attribute names, same buffers) so code that
reaches into ``render_loop.camera`` etc. keeps working.  Nothing here imports pynput /
pyautogui (the reference's control.py does, at import time: control.py:3-4,14).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from . import ops
from .compiler import compiled_for
from .rendering.ray_marching import PinholeCamera, SDFMarcher, SDFNormals
from .rendering.shader import Shader


class CapturedFrame:
    """One inference frame of a RenderLoop captured into a HIP graph (torch.cuda.CUDAGraph).

    A frame is 2-3 kernel launches (workspace init, k_render_fwd, optional k_shade_finish) plus a few
    hundred microseconds of Python; for small frames (main.py's 1440x900x32 takes ~0.15 ms of GPU time)
    the host side dominates.  Replaying a captured graph removes it.  Pose and scene parameters live in
    static device buffers that are refreshed (one tiny copy each) before every replay, so moving the camera
    or editing / optimising scene parameters needs no re-capture; changing mode, steps, resolution or the
    scene topology does.  Inference only (no autograd)."""

    def __init__(self, loop: "RenderLoop", mode: int = 0, degree: int = 1, marching_steps: int = 32, rows=None):
        self.loop, self.mode, self.degree, self.steps, self.rows = loop, mode % 8, int(degree), int(marching_steps), rows
        rp = loop._f32_buffer("ray_positions")
        rd = loop._f32_buffer("ray_directions")
        dev, n = rp.device, rp.shape[0]
        self.cs = compiled_for(loop.scene)
        self.q = torch.zeros(n, 4, dtype=torch.float32, device=dev)
        self.q[:, 0] = 1.0
        self.t = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        with torch.no_grad():
            self.params = self.cs.pack_params(dev).clone()
        cmap = loop._cmap_f32(dev) if self.mode in (6, 7) else None
        flags = ops.default_flags(loop.early_out, loop.tile8x8, loop.dynamic_tiles)

        def frame():
            return ops.render_frame(self.params, self.q, self.t, self.cs, rp, rd, loop.normals.tetra(), cmap,
                                    self.mode, self.degree, self.steps, rows, flags, None, loop.precision)

        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    frame()                       # warm-up outside capture (library / allocator state)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.image = frame()

    def __call__(self, orientations: Tensor, translations: Tensor) -> Tensor:
        """Render with the given pose; returns the graph's static output tensor [N,rows,W,3]
        (overwritten by the next replay -- clone it to keep it)."""
        with torch.no_grad():
            self.q.copy_(orientations)
            self.t.copy_(translations)
            fresh = self.cs.pack_params(self.params.device)
            if fresh.data_ptr() != getattr(self, "_seen", None):
                self.params.copy_(fresh)          # parameters changed since the last replay
                self._seen = fresh.data_ptr()
        self.graph.replay()
        return self.image


class RenderLoop(nn.Module):
    def __init__(self, scene, num_cameras: int = 1, px_width: int = 800, px_height: int = 800,
                 focal_length: float = 17e-3, sensor_width: float = 17e-3, sensor_height: float = 17e-3,
                 normals_eps: float = 5e-2, early_out: bool = True, tile8x8: bool = True,
                 dynamic_tiles: bool = True, precision: str = "exact"):
        super().__init__()
        self.scene = scene
        self.px_width = px_width
        self.px_height = px_height
        self.camera = PinholeCamera(num_cameras=num_cameras, px_width=px_width, px_height=px_height,
                                    focal_length=focal_length, sensor_width=sensor_width,
                                    sensor_height=sensor_height)
        self.marcher = SDFMarcher(sdf_scene=self.scene, early_out=early_out)
        self.normals = SDFNormals(sdf_scene=self.scene, normals_eps=normals_eps)
        self.shader = Shader()
        self.early_out = early_out
        self.tile8x8 = tile8x8
        self.dynamic_tiles = dynamic_tiles
        if precision not in ("exact", "fast"):
            raise ValueError("precision must be 'exact' (bit-faithful to the reference's CPU op stream, default) or 'fast'")
        self.precision = precision
        self._f32_cache = {}

    def _f32_buffer(self, name: str) -> Tensor:
        """Camera buffers as fp32 (the kernels' I/O type).  When the module was cast with
        .to(float16) the values are the fp16-rounded ones, like the reference's cast buffers."""
        buf = getattr(self.camera, name)
        if buf.dtype == torch.float32 and buf.is_contiguous():
            return buf
        key = (name, buf.data_ptr(), buf._version, buf.dtype)
        hit = self._f32_cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, buf.float().contiguous())
            self._f32_cache[name] = hit
        return hit[1]

    def _cmap_f32(self, device) -> Tensor:
        hit = self._f32_cache.get("cmap")
        cm = self.shader.cyclic_cmap
        key = (cm.data_ptr(), cm._version, str(device))
        if hit is None or hit[0] != key:
            hit = (key, cm.to(device=device, dtype=torch.float32).contiguous())
            self._f32_cache["cmap"] = hit
        return hit[1]

    def capture(self, mode: int = 0, degree: int = 1, marching_steps: int = 32, rows=None) -> CapturedFrame:
        """HIP-graph replay of one inference frame (see CapturedFrame)."""
        return CapturedFrame(self, mode, degree, marching_steps, rows)

    def forward(self, orientations: Tensor, translations: Tensor, mode: int = 0, degree: int = 1,
                marching_steps: int = 32, rows=None, allreduce_minmax=None):
        """-> image [N, H, W, 3].  ``rows=(r0, r1)`` renders only that pixel-row band
        ([N, r1-r0, W, 3]); ``allreduce_minmax`` is the hook row-tiled multi-GPU rendering uses
        for the global min/max of modes 1/2/5 (see ray_marching_amd/distributed.py)."""
        mode = mode % 8
        rp = self._f32_buffer("ray_positions")
        rd = self._f32_buffer("ray_directions")
        cs = compiled_for(self.scene)
        cmap = self._cmap_f32(rp.device) if mode in (6, 7) else None
        image = ops.render_frame(cs.pack_params(rp.device), orientations, translations, cs, rp, rd,
                                 self.normals.tetra(), cmap, mode, int(degree), int(marching_steps), rows,
                                 ops.default_flags(self.early_out, self.tile8x8, self.dynamic_tiles), allreduce_minmax,
                                 self.precision)
        if mode in (6, 7) and self.shader.cyclic_cmap.dtype == torch.float64:
            return image.double()  # reference: fp32 brightness * float64 colormap -> float64 image
        out_dtype = self.camera.ray_positions.dtype
        return image if out_dtype == torch.float32 else image.to(out_dtype)
