#!/usr/bin/env python3
"""The reference's main.py (1440x900, make_test_scene2, 32 steps, camera at (0,0,1)) on a headless
MI355X: scripted camera path instead of the mouse/keyboard, pinned-host frame sink instead of the GL
window.   python examples/headless_main.py --frames 300 [--out frames/]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ray_marching_amd.control import RenderLoop  # noqa: E402
from ray_marching_amd.headless import FrameSink, PosePlayer, run_headless  # noqa: E402
from ray_marching_amd.scene.scene_registry import make_test_scene2  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--out", default=None)
    ap.add_argument("--dtype", default="float32", choices=["float32", "float16"])
    ap.add_argument("--graph", action="store_true", help="replay each shader mode's frame from a captured HIP graph")
    a = ap.parse_args()
    device, dtype = torch.device("cuda"), getattr(torch, a.dtype)
    num_cameras, (px_width, px_height), px_size, marching_steps = 1, (1440, 900), 3.45e-6, 32   # main.py:20-26
    scene = make_test_scene2().to(device, dtype)
    render_loop = RenderLoop(scene=scene, num_cameras=num_cameras, px_width=px_width, px_height=px_height,
                             focal_length=px_size * px_height, sensor_width=px_size * px_width,
                             sensor_height=px_size * px_height, normals_eps=5e-2).to(device, dtype)
    events = PosePlayer(initial_position=[(0.0, 0.0, 1.0)], initial_orientation=[(1.0, 0.0, 0.0, 0.0)],
                        marching_steps=marching_steps, velocity=(0.0, 0.0, -0.05), angular_velocity=(0.0, 0.02, 0.0),
                        mode_every=40, device=device)
    events.position, events.orientation = events.position.to(dtype), events.orientation.to(dtype)
    events.velocity, events.angular_velocity = events.velocity.to(dtype), events.angular_velocity.to(dtype)
    window = FrameSink(px_width, px_height, "Window", out_dir=a.out)
    if a.graph:
        frames = {}

        def graphed(orientations, positions, mode, degree, steps):
            key = (mode % 8, degree, steps)
            if key not in frames:
                frames[key] = render_loop.capture(mode, degree, steps)
            return frames[key](orientations, positions)
        fps = run_headless(graphed, events, window, a.frames)
    else:
        fps = run_headless(render_loop, events, window, a.frames)
    print(f"{fps:.1f} frames per second ({px_width}x{px_height}, {marching_steps} steps, {a.dtype}, all 8 shader modes cycled)")
