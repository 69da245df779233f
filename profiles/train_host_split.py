"""Host time of the parts of an eager config-4 training step (perf_counter around each part, GPU never waited for)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd import ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_closed_test_scene
dev = torch.device("cuda:0")
h = w = 512
scene = make_closed_test_scene()
loop = RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                  sensor_height=bench.PX * h, normals_eps=bench.EPS).to(dev)
q = torch.tensor([[1.0, 0, 0, 0]], device=dev); t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
target = torch.rand(1, h, w, 1, device=dev)
params = list(scene.parameters())
acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return wrap


ops.Render.backward = staticmethod(timed("  Render.backward (python)", ops.Render.backward))
ops.Render.run = staticmethod(timed("  Render.run (python + launch)", ops.Render.run))
N = 300
for phase in ("warm", "timed"):
    acc.clear()
    if phase == "timed":
        torch.cuda.synchronize()
    T0 = time.perf_counter()
    for _ in range(20 if phase == "warm" else N):
        t0 = time.perf_counter()
        for p in params:
            p.grad = None
        t1 = time.perf_counter()
        img = loop(q, t, 0, 1, 64)
        t2 = time.perf_counter()
        loss = (img[..., :1] - target).pow(2).mean()
        t3 = time.perf_counter()
        loss.backward()
        t4 = time.perf_counter()
        for k, v in (("zero grads", t1 - t0), ("forward call", t2 - t1), ("loss ops", t3 - t2), ("backward call", t4 - t3)):
            acc[k] = acc.get(k, 0.0) + v
    total = time.perf_counter() - T0
    torch.cuda.synchronize()
print(f"host loop {1e3 * total / N:.3f} ms/step (GPU work per step ~0.40 ms)")
for k, v in acc.items():
    print(f"  {k:34s} {1e6 * v / N:7.1f} us")
