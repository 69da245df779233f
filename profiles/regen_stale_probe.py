"""How fast the per-ray dealing order of the regeneration kernels goes stale: the order is recorded at pose A, frames are
timed at pose A + delta with that order (adaptive_order huge: never refreshed) and with a fresh one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
dev = torch.device("cuda:0")
h, w = 1080, 1920
q = torch.tensor([[1.0, 0, 0, 0]], device=dev)


def mk(**kw):
    return RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                      sensor_height=bench.PX * h, normals_eps=bench.EPS, **kw).to(dev)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return 1e3 * best


tile = mk(adaptive_order=0, regen=False)
PER_RAY = len(sys.argv) > 1 and sys.argv[1] == "ray"
print("dealing order per", "ray" if PER_RAY else "tile (score from the ray costs)")
_mk = mk
def mk(**kw):
    if kw.get("regen"):
        kw["order_per_ray"] = PER_RAY
    return _mk(**kw)
for z in (-3.0, 1.0):
    print(f"camera (0,0,{z:+g})")
    for dx in (0.0, 0.003, 0.03, 0.3):
        stale = mk(regen=True, adaptive_order=1 << 30)
        fresh = mk(regen=True, adaptive_order=1 << 30)
        tA = torch.tensor([[0.0, 0.0, z]], device=dev)
        tB = torch.tensor([[dx, 0.0, z]], device=dev)
        with torch.no_grad():
            stale(q, tA, 4, 1, 128); fresh(q, tB, 4, 1, 128)        # frame 1 of each loop records its order
            a, b, c = timeit(lambda: stale(q, tB, 4, 1, 128)), timeit(lambda: fresh(q, tB, 4, 1, 128)), timeit(lambda: tile(q, tB, 4, 1, 128))
            same = torch.equal(stale(q, tB, 4, 1, 128), tile(q, tB, 4, 1, 128))
        print(f"  moved by {dx:5.3f} (x): order of the old pose {a:6.1f} us, fresh order {b:6.1f} us, tile kernel {c:6.1f} us, same image {same}", flush=True)
    # a rotation of the camera about y by small angles
    for deg in (0.5, 10.0):
        import math
        ang = math.radians(deg) / 2
        qB = torch.tensor([[math.cos(ang), 0.0, math.sin(ang), 0.0]], device=dev)
        stale = mk(regen=True, adaptive_order=1 << 30)
        fresh = mk(regen=True, adaptive_order=1 << 30)
        tA = torch.tensor([[0.0, 0.0, z]], device=dev)
        with torch.no_grad():
            stale(q, tA, 4, 1, 128); fresh(qB, tA, 4, 1, 128)
            a, b, c = timeit(lambda: stale(qB, tA, 4, 1, 128)), timeit(lambda: fresh(qB, tA, 4, 1, 128)), timeit(lambda: tile(qB, tA, 4, 1, 128))
        print(f"  turned by {deg:4.1f} deg: order of the old pose {a:6.1f} us, fresh order {b:6.1f} us, tile kernel {c:6.1f} us", flush=True)
