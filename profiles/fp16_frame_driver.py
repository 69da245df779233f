"""Kernel list of RenderLoop.to(float16) frames (BASELINE configs[2] shape): run under rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H = 3.45e-6, 3840, 2160
dev = torch.device("cuda:0")
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX * H, sensor_width=PX * W,
                  sensor_height=PX * H, normals_eps=5e-2).to(dev).to(torch.float16)
q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev, dtype=torch.float16)
t = torch.tensor([[0.0, 0.0, -3.0]], device=dev, dtype=torch.float16)
with torch.no_grad():
    for i in range(12):
        img = loop(q, t, (4, 0)[i % 2], 1, 256)
torch.cuda.synchronize()
print(img.dtype, img.shape)
