"""Frames for rocprofv3: config-2 frame with ray regeneration, camera z from argv (default 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
z = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
regen = {"0": False, "1": True, "auto": "auto"}[sys.argv[2]] if len(sys.argv) > 2 else True
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dev = torch.device("cuda:0")
h, w = 1080, 1920
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                  sensor_height=bench.PX * h, normals_eps=bench.EPS, regen=regen, adaptive_order=0 if regen is False else 16).to(dev)
q = torch.tensor([[1.0, 0, 0, 0]], device=dev)
t = torch.tensor([[0.0, 0.0, z]], device=dev)
with torch.no_grad():
    for i in range(frames):
        loop(q, t, (4, 0)[i % 2], 1, 128)
torch.cuda.synchronize()
