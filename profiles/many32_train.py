"""fwd+bwd of a Lambertian MSE step on the 32-primitive scene at 256^2 x 64 (specialised forward, generic backward):
for rocprofv3 --kernel-trace --stats.   python profiles/many32_train.py [n_prims] [size]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
scene = make_many_primitive_scene(n)
loop = RenderLoop(scene, num_cameras=1, px_width=size, px_height=size, focal_length=bench.PX*size, sensor_width=bench.PX*size, sensor_height=bench.PX*size, normals_eps=bench.EPS).to(dev)
q = torch.tensor([[1.0,0,0,0]], device=dev); t = torch.tensor([[0.0,0.0,-4.0]], device=dev)
target = torch.rand(1, size, size, 1, generator=torch.Generator().manual_seed(1)).to(dev)
def step():
    for p in scene.parameters(): p.grad = None
    (loop(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean().backward()
step(); step(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize()
print(f"many{n} {size}x{size}x64 fwd+bwd {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms/step")
