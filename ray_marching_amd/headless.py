"""Running the reference's frame loop (main.py:53-88) on a display-less MI355X node.

The reference feeds ``RenderLoop`` from ``EventAggregator.get_state()`` (pynput / pyautogui, needs an X
display; control.py:114-176) and shows the result with ``torchwindow.Window.draw`` (SDL2 + OpenGL +
CUDA-GL interop; torchwindow/window.py:146-174).  Neither exists on an accelerator node.  This module
supplies stand-ins with the same contracts:

* ``PosePlayer.get_state()`` returns the same 6-tuple ``(positions[N,3], orientations[N,4], mode,
  degree, marching_steps, save_frame)`` from a scripted camera path, integrating poses the way the
  reference does (control.py:150-165): ``position += rotate(dx * 0.1, q)``,
  ``q <- normalize(q (x) versor(dtheta * 0.25))``.
* ``FrameSink.draw(image)`` accepts exactly what ``Window.draw`` accepts -- a contiguous device tensor
  ``[H, W, 4]`` fp32 (row pitch 16*W bytes) -- and copies it to pinned host memory on a side stream
  (optionally writing binary PPM files) instead of a GL texture.

Host-side plumbing only; nothing here is on the data-parallel path.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import quaternion as Q


class PosePlayer:
    """Scripted replacement of EventAggregator: a constant body-frame velocity / angular velocity
    per frame (the same Lie-algebra update the key bindings drive), plus optional mode cycling."""

    translation_sensitivity = 0.1     # control.py:29-30
    rotation_sensitivity = 0.25

    def __init__(self, initial_position, initial_orientation, marching_steps: int = 32, mode: int = 0,
                 degree: int = 2, velocity=(0.0, 0.0, 0.0), angular_velocity=(0.0, 0.0, 0.0),
                 mode_every: int = 0, device="cpu"):
        self.position = torch.tensor(initial_position, dtype=torch.float32, device=device)
        self.orientation = torch.tensor(initial_orientation, dtype=torch.float32, device=device)
        self.velocity = torch.tensor([velocity], dtype=torch.float32, device=device)
        self.angular_velocity = torch.tensor([angular_velocity], dtype=torch.float32, device=device)
        self.marching_steps, self.mode, self.degree = marching_steps, mode, degree
        self.mode_every = mode_every
        self.frame = 0
        self.running = True

    def to(self, device):
        for name in ("position", "orientation", "velocity", "angular_velocity"):
            setattr(self, name, getattr(self, name).to(device))
        return self

    def get_state(self):
        self.position = Q.rotation((self.velocity * self.translation_sensitivity).expand_as(self.position),
                                   self.orientation) + self.position
        step = Q.to_versor(self.angular_velocity * self.rotation_sensitivity).expand_as(self.orientation)
        self.orientation = F.normalize(Q.multiply(self.orientation, step), p=2, dim=-1, eps=0)
        self.frame += 1
        if self.mode_every and self.frame % self.mode_every == 0:
            self.mode += 1
        return (self.position, self.orientation, self.mode, self.degree, self.marching_steps, False)


def to_rgba(images: torch.Tensor) -> torch.Tensor:
    """What main.py:78-84 hands to Window.draw: mean over cameras, fp32, alpha = 1 -> [H, W, 4].  (For one camera
    RenderLoop.display_frame lets the frame kernel write this tensor itself.)"""
    return F.pad(images.mean(dim=0).float(), pad=[0, 1], value=1.0)


class FrameSink:
    """Headless Window: ``draw([H,W,4] fp32 device tensor)`` -> pinned host ring buffer (+ one PPM / PNG / raw float32
    file per frame when ``out_dir`` is given)."""

    def __init__(self, width: int, height: int, name: str = "Window", out_dir: Optional[str] = None, ring: int = 2,
                 file_format: str = "ppm"):
        if file_format not in ("ppm", "png", "raw"):
            raise ValueError("file_format must be 'ppm', 'png' or 'raw' (the [H,W,4] float32 frame as is)")
        self.width, self.height, self.name = width, height, name
        self.out_dir, self.file_format = out_dir, file_format
        if out_dir:
            os.makedirs(out_dir, exist_ok=True)
        self._host = [torch.empty((height, width, 4), dtype=torch.float32).pin_memory() if torch.cuda.is_available()
                      else torch.empty((height, width, 4), dtype=torch.float32) for _ in range(ring)]
        self._events = [None] * ring
        self._stream = None
        self.frames = 0

    def draw(self, tensor: torch.Tensor):
        # the checks Window.draw relies on implicitly (cudaMemcpy2DToArrayAsync with pitch 16*W)
        if tensor.shape != (self.height, self.width, 4):
            raise ValueError(f"draw() expects [{self.height},{self.width},4], got {tuple(tensor.shape)}")
        if tensor.dtype != torch.float32 or not tensor.is_contiguous():
            raise ValueError("draw() expects a contiguous float32 tensor (row pitch 16*W bytes)")
        slot = self.frames % len(self._host)
        if tensor.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=tensor.device)
            if self._events[slot] is not None:
                self._events[slot].synchronize()             # the slot's previous copy has landed
            self._stream.wait_stream(torch.cuda.current_stream(tensor.device))
            with torch.cuda.stream(self._stream):
                self._host[slot].copy_(tensor, non_blocking=True)
                tensor.record_stream(self._stream)
                ev = torch.cuda.Event()
                ev.record(self._stream)
            self._events[slot] = ev
        else:
            self._host[slot].copy_(tensor)
        if self.out_dir:
            path = os.path.join(self.out_dir, f"frame_{self.frames:05d}.{self.file_format}")
            {"ppm": self.save_ppm, "png": self.save_png, "raw": self.save_raw}[self.file_format](path, slot)
        self.frames += 1

    def latest(self) -> torch.Tensor:
        """Most recent frame on the host (waits for its copy)."""
        slot = (self.frames - 1) % len(self._host)
        if self._events[slot] is not None:
            self._events[slot].synchronize()
        return self._host[slot]

    def _host_frame(self, slot: Optional[int]) -> torch.Tensor:
        if slot is None:
            return self.latest()
        if self._events[slot] is not None:
            self._events[slot].synchronize()
        return self._host[slot]

    def _rgb8(self, slot: Optional[int]) -> torch.Tensor:
        return (torch.nan_to_num(self._host_frame(slot)[..., :3]).clamp(0, 1) * 255.0 + 0.5).to(torch.uint8).contiguous()

    def save_ppm(self, path: str, slot: Optional[int] = None):
        rgb = self._rgb8(slot)
        with open(path, "wb") as f:
            f.write(f"P6 {self.width} {self.height} 255\n".encode())
            f.write(rgb.numpy().tobytes())

    def save_png(self, path: str, slot: Optional[int] = None):
        """8-bit RGB PNG (zlib from the standard library: one IHDR, one IDAT, filter type 0 on every row)."""
        import struct
        import zlib
        rows = self._rgb8(slot).numpy()
        raw = b"".join(b"\x00" + rows[y].tobytes() for y in range(self.height))

        def chunk(tag: bytes, data: bytes) -> bytes:
            return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

        with open(path, "wb") as f:
            f.write(b"\x89PNG\r\n\x1a\n")
            f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", self.width, self.height, 8, 2, 0, 0, 0)))
            f.write(chunk(b"IDAT", zlib.compress(raw, 6)))
            f.write(chunk(b"IEND", b""))

    def save_raw(self, path: str, slot: Optional[int] = None):
        """The frame exactly as Window.draw received it: H*W*4 little-endian float32 (NaNs and all)."""
        with open(path, "wb") as f:
            f.write(self._host_frame(slot).numpy().tobytes())

    def close(self):
        for ev in self._events:
            if ev is not None:
                ev.synchronize()


def run_headless(render_loop, events, window: FrameSink, max_frames: int, fused_display: bool = False):
    """The body of main.py:53-88 with the stand-ins above.  Returns frames per second."""
    import time
    t0 = time.time()
    n = 0
    with torch.no_grad():
        while events.running and n < max_frames:
            positions, orientations, mode, degree, marching_steps, _ = events.get_state()
            if fused_display and orientations.shape[0] == 1:      # the frame kernel writes the [H,W,4] display tensor itself
                window.draw(render_loop.display_frame(orientations, positions, mode, degree, marching_steps))
            else:
                images = render_loop(orientations, positions, mode, degree, marching_steps)
                window.draw(to_rgba(images))
            n += 1
    window.close()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    return n / max(time.time() - t0, 1e-9)
