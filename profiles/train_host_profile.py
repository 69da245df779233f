"""Where the host time of an eager config-4 training step goes (cProfile over 300 steps, no synchronisation inside)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
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


def step():
    for p in params:
        p.grad = None
    (loop(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean().backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300):
    step()
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"300 steps: host loop {1e3 * host / 300:.3f} ms/step, with the final synchronise {1e3 * (time.perf_counter() - t0) / 300:.3f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
st.sort_stats("tottime").print_stats(18)
