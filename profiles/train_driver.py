"""Driver for rocprofv3 passes over the config-4 training step (closed make_test_scene, Lambertian MSE, 64 steps):

    python3 profiles/train_driver.py [size=512] [steps=12] [shader mode=0]

A few k_camera_fwd launches first (calibration of FETCH_SIZE / WRITE_SIZE: 2 x size^2 x 12 B read and written), then
`steps` eager training steps without an optimiser."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ray_marching_amd.control import RenderLoop  # noqa: E402
from ray_marching_amd.scene.scene_registry import make_closed_test_scene  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda:0")
scene = make_closed_test_scene()
loop = RenderLoop(scene, num_cameras=1, px_width=size, px_height=size, focal_length=bench.PX * size,
                  sensor_width=bench.PX * size, sensor_height=bench.PX * size, normals_eps=bench.EPS).to(dev)
q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
target = torch.rand(1, size, size, 1, device=dev)
with torch.no_grad():
    for _ in range(4):
        loop.camera(q, t)
for _ in range(steps):
    for p in scene.parameters():
        p.grad = None
    (loop(q, t, mode, 1, 64)[..., :1] - target).pow(2).mean().backward()
torch.cuda.synchronize()
print("train_driver done", size, steps)
