#!/usr/bin/env python3
"""Render a small strip of shader modes of make_test_scene2 with the HIP frame kernel and write docs/gallery.png
(tiny built-in PNG writer; no imaging dependency)."""
import os
import struct
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ray_marching_amd.control import RenderLoop  # noqa: E402
from ray_marching_amd.scene.scene_registry import make_test_scene2  # noqa: E402


def write_png(path, rgb8):
    h, w, _ = rgb8.shape
    raw = b"".join(b"\x00" + rgb8[y].tobytes() for y in range(h))

    def chunk(tag, data):
        c = struct.pack(">I", len(data)) + tag + data
        return c + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


if __name__ == "__main__":
    px, w, h = 3.45e-6, 480, 300
    dev = torch.device("cuda")
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=px * h * 0.9,
                      sensor_width=px * w, sensor_height=px * h, normals_eps=5e-2).to(dev)
    q = torch.nn.functional.normalize(torch.tensor([[0.97, -0.12, 0.18, 0.02]]), dim=-1).to(dev)
    t = torch.tensor([[-1.1, -0.9, -2.6]], device=dev)
    tiles = []
    with torch.no_grad():
        for mode in (0, 4, 6, 1):
            img = loop(q, t, mode, 2, 128)[0].float()
            tiles.append((torch.nan_to_num(img).clamp(0, 1) * 255 + 0.5).to(torch.uint8).cpu())
    strip = torch.cat(tiles, dim=1).numpy()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "docs", "gallery.png")
    write_png(out, strip)
    print(out, strip.shape)
