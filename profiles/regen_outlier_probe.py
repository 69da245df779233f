"""Per-frame GPU and host times of many consecutive regeneration frames: looks for rare slow frames."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd import ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
dev = torch.device("cuda:0")
h, w = 1080, 1920
q = torch.tensor([[1.0, 0, 0, 0]], device=dev)
for trial in range(3):
    for kw in (dict(regen=True), dict(regen=True, order_per_ray=True), dict()):
        loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                          sensor_height=bench.PX * h, normals_eps=bench.EPS, **kw).to(dev)
        t = torch.tensor([[0.0, 0.0, 1.0]], device=dev)
        n = 300
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        host = []
        with torch.no_grad():
            for i in range(20):
                loop(q, t, (4, 0)[i % 2], 1, 128)
            torch.cuda.synchronize()
            evs[0].record()
            t0 = time.perf_counter()
            for i in range(n):
                h0 = time.perf_counter()
                loop(q, t, (4, 0)[i % 2], 1, 128)
                host.append(time.perf_counter() - h0)
                evs[i + 1].record()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        gpu = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
        srt = sorted(gpu)
        slow = [(i, round(g, 3)) for i, g in enumerate(gpu) if g > 2 * srt[n // 2]]
        print(f"trial {trial} {kw}: wall {1e3 * wall / n:.3f} ms/frame; GPU median {srt[n // 2]:.3f} p99 {srt[int(n * .99)]:.3f} max {srt[-1]:.3f}; "
              f"host max {1e3 * max(host):.2f} ms median {1e3 * sorted(host)[n // 2]:.3f}; frames > 2x median: {slow[:12]}", flush=True)
        del loop
