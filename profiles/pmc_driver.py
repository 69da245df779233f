"""Small driver for rocprofv3 --pmc passes: a calibration kernel with a known byte count
(k_camera_fwd: reads 2 x [1,1080,1920,3] fp32, writes 2 x the same, with the same 12-byte
per-lane access pattern the frame kernel uses) followed by the config-2 frame kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ray_marching_amd.control import RenderLoop  # noqa: E402
from ray_marching_amd.scene.scene_registry import make_test_scene2  # noqa: E402

PX, W, H = 3.45e-6, 1920, 1080
dev = torch.device("cuda:0")
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX * H,
                  sensor_width=PX * W, sensor_height=PX * H, normals_eps=5e-2, regen=False).to(dev)   # the tile kernel only
q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
t = torch.tensor([[0.0, 0.0, float(sys.argv[1]) if len(sys.argv) > 1 else -3.0]], device=dev)     # camera z (default -3)
with torch.no_grad():
    for _ in range(4):
        loop.camera(q, t)                     # calibration: 49.8 MB read, 49.8 MB written
    for _ in range(4):
        loop(q, t, 4, 1, 128)                 # k_render_fwd, normal shader
    for _ in range(4):
        loop(q, t, 0, 1, 128)                 # k_render_fwd, lambertian shader
torch.cuda.synchronize()
