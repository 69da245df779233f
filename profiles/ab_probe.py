"""Same-box A/B of compile-time variants (RM_HIPCC_EXTRA) and host knobs: config-2 frame at both cameras, config-4
fwd+bwd step (eager loop, no sync inside), config-5 tile.  Each variant runs in its own process and builds its own
libraries (hipcc on the box).  usage: python profiles/ab_probe.py "name=-DFLAG ..." "name2=ENV:RM_AB_PACK=1" ..."""
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
def mk(scene, h, w):
    return RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX*h, sensor_width=bench.PX*w, sensor_height=bench.PX*h, normals_eps=bench.EPS).to(dev)
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
loop = mk(make_test_scene2(), 1080, 1920)
for z in (-3.0, 1.0):
    t = torch.tensor([[0.0,0.0,z]], device=dev)
    with torch.no_grad():
        out["c2_z%%g_us" %% z] = round(1e3 * timeit(lambda: loop(q, t, 4, 1, 128), 30), 1)
scene = make_closed_test_scene(); loop4 = mk(scene, 512, 512)
t = torch.tensor([[0.0,0.0,-1.0]], device=dev); target = torch.rand(1,512,512,1, device=dev)
def step():
    for p in scene.parameters(): p.grad = None
    (loop4(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean().backward()
out["c4_fwdbwd_us"] = round(1e3 * timeit(step, 30), 1)
with torch.no_grad():
    out["c4_fwd_nograd_us"] = round(1e3 * timeit(lambda: loop4(q, t, 0, 1, 64), 30), 1)
loop5 = mk(make_many_primitive_scene(32), 4320, 7680)
t5 = torch.tensor([[0.0,0.0,-4.5]], device=dev)
with torch.no_grad():
    out["c5_tile_ms"] = round(timeit(lambda: loop5(q, t5, 4, 1, 256, rows=(1620, 2160)), 3), 2)
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
    # every variant builds into a directory of its own: the product library is never replaced (ray_marching_amd/_build.py)
    env["RM_LIB_DIR"] = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rm_ab_" + "".join(c if c.isalnum() else "_" for c in name))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"{name:28s} {res[0][7:] if res else 'FAILED: ' + r.stderr[-400:]}   ({time.time()-t0:.0f} s)", flush=True)

if __name__ == "__main__":
    for arg in sys.argv[1:]:
        name, _, spec = arg.partition("=")
        run(name, spec)
