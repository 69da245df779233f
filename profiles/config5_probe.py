"""BASELINE config 5 on ONE GPU: one of the 8 row tiles (7680 x 540 rows of the 7680x4320 frame),
256 march steps, 32-primitive smooth-union scene, normal shader."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.compiler import compiled_for
from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
PX, W, H, S = 3.45e-6, 7680, 4320, 256
dev = torch.device("cuda:0")
loop = RenderLoop(make_many_primitive_scene(32), num_cameras=1, px_width=W, px_height=H, focal_length=PX * H,
                  sensor_width=PX * W, sensor_height=PX * H, normals_eps=5e-2).to(dev)
print("specialised:", compiled_for(loop.scene).specialised)
q = torch.tensor([[1.0, 0, 0, 0]], device=dev); t = torch.tensor([[0.0, 0.0, -4.5]], device=dev)
for band in (3, 0):
    rows = (band * 540, (band + 1) * 540)
    with torch.no_grad():
        loop(q, t, 4, 1, S, rows=rows)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            img = loop(q, t, 4, 1, S, rows=rows)
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    rays = W * 540
    print(f"band {band}: {ms:.2f} ms/tile  {rays / ms / 1e3:.1f} Mrays/s  {rays * (S + 6) / ms / 1e6:.1f} G ray-evals/s  finite={torch.isfinite(img).all().item()}")
