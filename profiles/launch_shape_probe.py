import os, subprocess, sys, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for block, cap, extra in ((256, 2048, ""), (256, 2048, "--static-tiles"), (256, 0, "--static-tiles"), (128, 0, "--static-tiles"), (64, 0, "--static-tiles"),
                          (256, 1024, ""), (256, 4096, ""), (64, 8192, "")):
    env = dict(os.environ, RM_BLOCK=str(block), RM_MAX_BLOCKS=str(cap))
    for z in ("-3.0", "1.0"):
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "3", "--no-cpu-baseline",
                              "--skip-backward", "--camera-z", z] + extra.split(), env=env, capture_output=True, text=True).stdout
        d = json.loads(out)
        print(f"block {block:3d} cap {cap:5d} {extra or 'dynamic':15s} z={z:5s}: {d['value']:7.1f} Mrays/s  kernel {d['roofline']['kernel_ms']*1e3:6.1f} us", flush=True)
