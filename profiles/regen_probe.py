"""Ray regeneration (RM_FLAG_REGEN) against the tile kernel on one box: identical images, time per frame at both
poses of config 2, the 512^2 closed scene and one config-5 band.  Compile-time variants through RM_HIPCC_EXTRA.
usage: python profiles/regen_probe.py ["name=-DRM_REGEN_MIN_FREE=8" ...]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2, make_closed_test_scene, make_many_primitive_scene
dev = torch.device("cuda:0")
def mk(scene, h, w, **kw):
    return RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX*h, sensor_width=bench.PX*w, sensor_height=bench.PX*h, normals_eps=bench.EPS, **kw).to(dev)
def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
out = {}
q = torch.tensor([[1.0,0,0,0]], device=dev)
def pair(name, scene, h, w, t, mode, steps, rows=None, reps=32):
    loops = {"tile": mk(scene, h, w, regen=False), "regen": mk(scene, h, w, regen=True),
             "regen_per_ray": mk(scene, h, w, regen=True, order_per_ray=True), "auto": mk(scene, h, w)}
    res = {}
    with torch.no_grad():
        ref = loops["tile"](q, t, mode, 1, steps, rows=rows)
        for k, l in loops.items():
            img = l(q, t, mode, 1, steps, rows=rows)
            same = bool(torch.equal(img.view(torch.int32), ref.view(torch.int32)))
            res[k + "_us"] = round(1e3 * timeit(lambda: l(q, t, mode, 1, steps, rows=rows), reps), 1)
            if not same:
                res[k + "_DIFFERS"] = True
    out[name] = res
s2 = make_test_scene2()
for z in (-3.0, 1.0):
    pair("c2_z%%g" %% z, s2, 1080, 1920, torch.tensor([[0.0,0.0,z]], device=dev), 4, 128)
pair("c2_z1_mode1", s2, 1080, 1920, torch.tensor([[0.0,0.0,1.0]], device=dev), 1, 128)
pair("c4_fwd", make_closed_test_scene(), 512, 512, torch.tensor([[0.0,0.0,-1.0]], device=dev), 0, 64)
pair("c5_band", make_many_primitive_scene(32), 4320, 7680, torch.tensor([[0.0,0.0,-4.5]], device=dev), 4, 256, rows=(1620, 2160), reps=16)
print("RESULT " + json.dumps(out))
''' % ROOT

def run(name, spec):
    env = dict(os.environ, RM_SPECIALIZE="jit")
    flags = []
    for tok in spec.split():
        if tok.startswith("ENV:"):
            k, v = tok[4:].split("=", 1); env[k] = v
        else:
            flags.append(tok)
    env["RM_HIPCC_EXTRA"] = " ".join(flags)
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"{name:24s} {res[0][7:] if res else 'FAILED: ' + r.stderr[-1500:]}   ({time.time()-t0:.0f} s)", flush=True)

if __name__ == "__main__":
    for arg in (sys.argv[1:] or ["default="]):
        name, _, spec = arg.partition("=")
        run(name, spec)
