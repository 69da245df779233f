"""BASELINE config 3: make_test_scene2, 3840x2160, 256 march steps, fp16 module (fp16 I/O, fp32 arithmetic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H, S = 3.45e-6, 3840, 2160, 256
dev = torch.device("cuda:0")
for dtype in (torch.float16, torch.float32):
    loop = RenderLoop(make_test_scene2().to(dev, dtype), num_cameras=1, px_width=W, px_height=H, focal_length=PX * H,
                      sensor_width=PX * W, sensor_height=PX * H, normals_eps=5e-2).to(dev, dtype)
    q = torch.tensor([[1.0, 0, 0, 0]], device=dev, dtype=dtype); t = torch.tensor([[0.0, 0.0, -3.0]], device=dev, dtype=dtype)
    for mode in (4, 0):
        with torch.no_grad():
            for _ in range(2): img = loop(q, t, mode, 1, S)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(5): img = loop(q, t, mode, 1, S)
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"{str(dtype):14s} mode {mode}: {ms:.3f} ms/frame  {W*H/ms/1e3:.0f} Mrays/s  {W*H*(S+6)/ms/1e6:.0f} G ray-evals/s  out {img.dtype} finite={torch.isfinite(img).all().item()}")
