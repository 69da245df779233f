"""A moving camera: 512 frames along a path from in front of the scene (0,0,-3) into the torus (0,0,1) and back, with
a slow turn -- the kind of input the reference's interactive loop produces.  Average frame time of RenderLoop's
regen="auto" against the fixed choices, and where auto switched.  By default every frame is waited for (the reference's
loop draws each frame before it reads the next pose); --free-running enqueues all 512 frames without waiting: the host
is then hundreds of frames ahead of the GPU and a host-side choice can only follow with that delay."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
dev = torch.device("cuda:0")
h, w, steps, n = 1080, 1920, 128, 512


def pose(i):
    s = 0.5 - 0.5 * math.cos(2 * math.pi * i / n)          # 0 -> 1 -> 0
    ang = math.radians(20.0 * math.sin(2 * math.pi * i / n)) / 2
    q = torch.tensor([[math.cos(ang), 0.0, math.sin(ang), 0.0]])
    t = torch.tensor([[0.3 * math.sin(4 * math.pi * i / n), 0.0, -3.0 + 4.0 * s]])
    return q, t


poses = [tuple(x.to(dev) for x in pose(i)) for i in range(n)]
SYNC = "--free-running" not in sys.argv      # default: wait for every frame, as main.py's window.draw does
ref = None
per_frame = {}
VARIANTS = (("tile kernel", dict(regen=False)), ("pool kernels", dict(regen=True)), ("pools, order every 4th frame", dict(regen=True, adaptive_order=4)),
            ("pools, natural order", dict(regen=True, adaptive_order=0)), ("auto", dict()))
for name, kw in VARIANTS:
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                      sensor_height=bench.PX * h, normals_eps=bench.EPS, **kw).to(dev)
    with torch.no_grad():
        for i in range(32):
            loop(*poses[0], 4, 1, steps)
        torch.cuda.synchronize()
        used, t0 = [], time.perf_counter()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        evs[0].record()
        for i in range(n):
            img = loop(*poses[i], 4, 1, steps)
            evs[i + 1].record()
            if SYNC:
                evs[i + 1].synchronize()       # an interactive loop consumes the frame before it knows the next pose
            if name == "auto":
                used.append(next(iter(loop._choice_state.values()))["regen"])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        check = loop(*poses[n // 2], 4, 1, steps)
    ref = check if ref is None else ref
    per_frame[name] = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
    line = f"{name:13s} {1e3 * dt:.4f} ms/frame = {h * w / dt / 1e9:.2f} Grays/s; frame {n // 2} identical to the tile kernel's: {bool(torch.equal(check, ref))}"
    if name == "auto":
        sw = [i for i in range(1, n) if used[i] != used[i - 1]]
        line += f"; pools in use for {sum(used)} of {n} frames, switches at frames {sw}"
        line += "\n   decisions (frame counter, pools in use, ms of the other kernel, ms of the kernel in use): " + str(next(iter(loop._choice_state.values()))["log"])
    print(line, flush=True)

print("mean GPU ms per frame over 32-frame segments of the path (camera z at the segment's middle):")
for a in range(0, n, 32):
    z = float(poses[a + 16][1][0, 2])
    print(f"  frames {a:3d}-{a + 31:3d} z={z:+.2f}: " + ", ".join(f"{k} {sum(v[a:a + 32]) / 32:.3f}" for k, v in per_frame.items()))
