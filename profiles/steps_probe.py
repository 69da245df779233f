"""k_render_fwd time vs march steps, early-out on/off (config-2 frame)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H = 3.45e-6, 1920, 1080
dev = torch.device("cuda:0")
q = torch.tensor([[1.0, 0, 0, 0]], device=dev); t = torch.tensor([[0.0, 0.0, -3.0]], device=dev)
def timeit(loop, S, mode=4, reps=10):
    with torch.no_grad():
        for _ in range(3): loop(q, t, mode, 1, S)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps): loop(q, t, mode, 1, S)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for early in (False, True):
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX*H, sensor_width=PX*W, sensor_height=PX*H, normals_eps=5e-2, early_out=early).to(dev)
    print("early" if early else "full ", " ".join(f"S={S}:{timeit(loop,S):.3f}" for S in (0, 8, 16, 32, 64, 128, 256)), " mode3(no normals) S=0:", f"{timeit(loop,0,3):.3f}")
