"""fwd+bwd step of BASELINE config 4 (closed make_test_scene, 512x512x64, Lambertian MSE) in a loop:
wall time per step next to the kernel durations (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_closed_test_scene
h = w = 512
scene = make_closed_test_scene()
loop = RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                  sensor_height=bench.PX * h, normals_eps=bench.EPS).to(dev)
q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev)
t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
target = torch.rand(1, h, w, 1, device=dev)
opt = torch.optim.Adam(scene.parameters(), lr=1e-3)
for reps in (5, 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        opt.zero_grad(set_to_none=True)
        loss = (loop(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean()
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    print(f"{reps} optimiser steps: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms/step (wall, incl. Adam)")
