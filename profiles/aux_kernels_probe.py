"""Achieved HBM bandwidth of the memory-bound auxiliary kernels (algorithmic bytes / time)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
from ray_marching_amd.scene.primitives import SDFSphere
PX, W, H = 3.45e-6, 3840, 2160
dev = torch.device("cuda:0")
def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX*H, sensor_width=PX*W, sensor_height=PX*H, normals_eps=5e-2).to(dev)
q = torch.tensor([[1.0, 0, 0, 0]], device=dev); t = torch.tensor([[0.0, 0.0, -3.0]], device=dev)
n = W * H
with torch.no_grad():
    dt = timeit(lambda: loop.camera(q, t)); print(f"k_camera_fwd  4K: {dt*1e6:7.1f} us  {n*48/dt/1e9:7.0f} GB/s (48 B/ray)")
    pts = torch.randn(n, 3, device=dev)
    for name, scene in (("sphere", SDFSphere(0.5).to(dev)), ("scene2", loop.scene)):
        dt = timeit(lambda: scene(pts)); print(f"k_sdf_fwd {name:7s}: {dt*1e6:7.1f} us  {n*16/dt/1e9:7.0f} GB/s (16 B/point)")
    pos, fr, _, dirs = loop.camera(q, t)
    nrm = torch.nn.functional.normalize(torch.randn(1, H, W, 3, device=dev), dim=-1)
    dt = timeit(lambda: loop.shader.lambertian_shader(dirs, nrm)); print(f"k_shade_fwd lambertian: {dt*1e6:7.1f} us  {n*36/dt/1e9:7.0f} GB/s (24 in + 12 out B/pixel)")
    dt = timeit(lambda: loop.normals(pos)); print(f"k_normals_fwd scene2: {dt*1e6:7.1f} us  {n*5/dt/1e9:7.1f} G evals/s")
    dt = timeit(lambda: loop.marcher(pos, dirs, 32)); print(f"k_march_fwd scene2 S=32: {dt*1e6:7.1f} us  {n/dt/1e6:7.0f} Mrays/s")
