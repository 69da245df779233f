"""A/B of the generic backward's accumulator layout (columns per thread / one row per wave, RM_WIDE_ACC_MIN) through the
LDS interpreter: fwd+bwd of a Lambertian MSE step, eager, ms per step.   python profiles/wide_ab.py 1000000 256 64 0"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_closed_test_scene, make_many_primitive_scene, make_test_scene2
dev = torch.device("cuda:0")
out = {}
for name, make, size, z in (("closed1", make_closed_test_scene, 512, -1.0), ("scene2", make_test_scene2, 512, -3.0),
                            ("many16", lambda: make_many_primitive_scene(16), 256, -4.0), ("many32", lambda: make_many_primitive_scene(32), 256, -4.0)):
    scene = make()
    loop = RenderLoop(scene, num_cameras=1, px_width=size, px_height=size, focal_length=bench.PX*size, sensor_width=bench.PX*size, sensor_height=bench.PX*size, normals_eps=bench.EPS).to(dev)
    q = torch.tensor([[1.0,0,0,0]], device=dev); t = torch.tensor([[0.0,0.0,z]], device=dev)
    target = torch.rand(1, size, size, 1, generator=torch.Generator().manual_seed(1)).to(dev)
    def step():
        for p in scene.parameters(): p.grad = None
        (loop(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean().backward()
    step(); step(); torch.cuda.synchronize()
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5): step()
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0) / 5 * 1e3)
    out[name + "_ms"] = round(sorted(runs)[1], 3)
    g = torch.cat([p.grad.flatten() for p in scene.parameters()])
    out[name + "_gsum"] = round(float(g.double().abs().sum()), 6)
print("RESULT " + json.dumps(out))
''' % ROOT
for thr in sys.argv[1:]:
    env = dict(os.environ, RM_SPECIALIZE="off", RM_HIPCC_EXTRA=f"-DRM_WIDE_ACC_MIN={thr}",
               RM_LIB_DIR=os.path.join(os.environ.get("TMPDIR", "/tmp"), f"rm_wide_{thr}"))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"RM_WIDE_ACC_MIN={thr:8s} {res[0][7:] if res else 'FAILED: ' + r.stderr[-600:]}   ({time.time()-t0:.0f} s)", flush=True)
