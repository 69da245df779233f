"""Same-box A/B of the config-4 training step (closed make_test_scene, Lambertian MSE, 64 steps) replayed from a HIP graph:
    python profiles/train_ab.py "name=-DFLAG ... ENV:RM_BWD_BLOCKS=1536" ...
Every variant runs in its own process, builds its libraries into its own directory (RM_LIB_DIR) and prints the replay
time per step at 512^2 and 1024^2 (median of 5 runs of 40 replays)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys, time, json
sys.path.insert(0, %r)
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.graphs import capture_step
from ray_marching_amd.scene.scene_registry import make_closed_test_scene
dev = torch.device("cuda:0")
out = {}
q = torch.tensor([[1.0,0,0,0]], device=dev); t = torch.tensor([[0.0,0.0,-1.0]], device=dev)
for size in (512, 1024):
    scene = make_closed_test_scene()
    loop = RenderLoop(scene, num_cameras=1, px_width=size, px_height=size, focal_length=bench.PX*size, sensor_width=bench.PX*size, sensor_height=bench.PX*size, normals_eps=bench.EPS, dynamic_tiles=os.environ.get('RM_AB_DYNAMIC', '1') == '1').to(dev)
    target = torch.rand(1, size, size, 1, device=dev)
    def step():
        (loop(q, t, 0, 1, 64)[..., :1] - target).pow(2).mean().backward()
    graph, _, _ = capture_step(step, list(scene.parameters()), warmup=3)
    graph.replay(); torch.cuda.synchronize()
    runs = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(40): graph.replay()
        torch.cuda.synchronize()
        runs.append((time.perf_counter() - t0) / 40 * 1e3)
    out["graph_ms_%%d" %% size] = round(sorted(runs)[2], 4)
    g = torch.cat([p.grad.flatten() for p in scene.parameters()])
    out["gsum_%%d" %% size] = float(g.double().abs().sum())
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
    env["RM_LIB_DIR"] = os.path.join(os.environ.get("TMPDIR", "/tmp"), "rm_ab_" + "".join(c if c.isalnum() else "_" for c in " ".join(flags) or "base"))
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    res = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"{name:34s} {res[0][7:] if res else 'FAILED: ' + r.stderr[-400:]}   ({time.time()-t0:.0f} s)", flush=True)

if __name__ == "__main__":
    for arg in sys.argv[1:]:
        name, _, spec = arg.partition("=")
        run(name, spec)
