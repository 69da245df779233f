"""PMC driver: one config-2 frame kernel, early-out off (fixed work), for instruction-class counters."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H = 3.45e-6, 1920, 1080
dev = torch.device("cuda:0")
q = torch.tensor([[1.0, 0, 0, 0]], device=dev); t = torch.tensor([[0.0, 0.0, -3.0]], device=dev)
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX*H, sensor_width=PX*W, sensor_height=PX*H, normals_eps=5e-2, early_out=False).to(dev)
with torch.no_grad():
    for _ in range(3): loop(q, t, 4, 1, 128)
torch.cuda.synchronize()
