import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for cap in (512, 768, 1024, 1280, 1536, 2048):
    env = dict(os.environ, RM_MAX_BLOCKS=str(cap))
    res = []
    for z in ("-3.0", "1.0"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "30", "--warmup", "3", "--no-cpu-baseline",
                              "--skip-backward", "--camera-z", z], env=env, capture_output=True, text=True).stdout
        d = json.loads(out); res.append(d['roofline']['kernel_ms'] * 1e3)
    print(f"cap {cap:5d}: z=-3 {res[0]:6.1f} us   z=+1 {res[1]:6.1f} us", flush=True)
