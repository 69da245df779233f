"""BASELINE configs[4] (7680x4320x256, 32-primitive scene, one GPU): the tile frame kernel against the ray pools.
    python profiles/config5_kernels.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
dev = torch.device("cuda:0")
W, H, S = 7680, 4320, 256
q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev); t = torch.tensor([[0.0, 0.0, -4.5]], device=dev)
ref = None
variants = (("tile kernel", dict(regen=False)), ("ray pools", dict(regen=True)), ("tile kernel again", dict(regen=False)))
if len(sys.argv) > 1 and sys.argv[1] == "pools":      # A/B of pool-kernel builds: RM_HIPCC_EXTRA=-D... RM_LIB_DIR=... RM_SPECIALIZE=jit
    variants = (("ray pools " + os.environ.get("RM_LABEL", os.environ.get("RM_HIPCC_EXTRA", "")), dict(regen=True)),)
for name, kw in variants:
    loop = RenderLoop(make_many_primitive_scene(32), num_cameras=1, px_width=W, px_height=H, focal_length=bench.PX * H,
                      sensor_width=bench.PX * W, sensor_height=bench.PX * H, normals_eps=bench.EPS, **kw).to(dev)
    with torch.no_grad():
        for _ in range(2):
            img = loop(q, t, 4, 1, S)
        torch.cuda.synchronize()
        times = []
        for _ in range(4):
            t0 = time.perf_counter(); img = loop(q, t, 4, 1, S); torch.cuda.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    same = "" if ref is None else f"  identical to the tile kernel's frame: {torch.equal(img, ref)}"
    if ref is None: ref = img.clone()
    print(f"{name:36s} {' '.join(f'{x:.1f}' for x in times)} ms{same}", flush=True)
    del loop, img
    torch.cuda.empty_cache()
