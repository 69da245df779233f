"""Driver for rocprofv3 passes over config 5 (32-primitive smooth union, 7680-wide, 256 steps, normal shader):

    python3 profiles/config5_driver.py [rows=540] [frames=3] [regen=auto|0|1]

k_camera_fwd calibration launches on a 1080p camera first, then `frames` renders of the band in the middle of the frame."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ray_marching_amd.control import RenderLoop  # noqa: E402
from ray_marching_amd.rendering.ray_marching import PinholeCamera  # noqa: E402
from ray_marching_amd.scene.scene_registry import make_many_primitive_scene  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 540
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3
regen = sys.argv[3] if len(sys.argv) > 3 else "0"
regen = "auto" if regen == "auto" else bool(int(regen))
PX, W, H, S = 3.45e-6, 7680, 4320, 256
dev = torch.device("cuda:0")
r0 = (H - rows) // 2 // 8 * 8
loop = RenderLoop(make_many_primitive_scene(32), num_cameras=1, px_width=W, px_height=H, focal_length=PX * H,
                  sensor_width=PX * W, sensor_height=PX * H, normals_eps=5e-2, rows=(r0, r0 + rows), regen=regen).to(dev)
cal = PinholeCamera(1, 1920, 1080, PX * 1080, PX * 1920, PX * 1080).to(dev)
q = torch.tensor([[1.0, 0, 0, 0]], device=dev)
t = torch.tensor([[0.0, 0.0, -4.5]], device=dev)
with torch.no_grad():
    for _ in range(4):
        cal(q, t)
    for _ in range(frames):
        loop(q, t, 4, 1, S)
torch.cuda.synchronize()
print("config5_driver done", rows, frames, regen)
