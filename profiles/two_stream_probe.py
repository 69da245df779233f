"""Do two independent frames on two HIP streams overlap their tails?  (config-2 frame, modes 4 and 0)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H, S = 3.45e-6, 1920, 1080, 128
dev = torch.device("cuda:0")
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX*H, sensor_width=PX*W, sensor_height=PX*H, normals_eps=5e-2).to(dev)
q = torch.tensor([[1.0,0,0,0]], device=dev); t = torch.tensor([[0.0,0.0,-3.0]], device=dev)
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
def run(nstreams, frames=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        for i in range(frames):
            if nstreams == 1:
                loop(q, t, (4, 0)[i & 1], 1, S)
            else:
                with torch.cuda.stream(streams[i % nstreams]):
                    loop(q, t, (4, 0)[i & 1], 1, S)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / frames * 1e3
for n in (1, 2, 3, 1, 2, 3):
    run(n, 10)
    print(f"{n} stream(s): {run(n):.4f} ms per frame  ({W*H/run(n)/1e3:.0f} Mrays/s)")
