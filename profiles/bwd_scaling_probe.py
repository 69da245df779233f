"""How do k_render_fwd (recording) / k_render_bwd scale with frame size and step count?  (run under
rocprofv3 --kernel-trace; the kernel_trace csv lists the launches in order: 12 per configuration)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_closed_test_scene
dev = torch.device("cuda:0")
for (h, steps) in ((512, 64), (512, 32), (1024, 64), (256, 64)):
    w = h
    scene = make_closed_test_scene()
    loop = RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                      sensor_height=bench.PX * h, normals_eps=bench.EPS).to(dev)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev); t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
    target = torch.rand(1, h, w, 1, device=dev)
    for _ in range(12):
        for p in scene.parameters(): p.grad = None
        (loop(q, t, 0, 1, steps)[..., :1] - target).pow(2).mean().backward()
    torch.cuda.synchronize()
    print("done", h, steps, flush=True)
