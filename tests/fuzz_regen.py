"""Fuzz of the ray-regeneration kernels: random scene trees (tests/helpers.random_spec: all 11 node types, nested
unions, un-normalised quaternions), random poses inside and outside the geometry, random frame sizes (mostly no
multiple of the 8x8 tile), step counts and shader modes, one or two cameras; k_march_regen + k_render_finish must
render the tile kernel's image bit for bit (NaN pixels included) -- in natural order and with the dealing orders made
from the recorded ray costs (tile scores, per-ray), through the interpreter.
    python tests/fuzz_regen.py [n_seeds]        (a script, not collected by pytest: ~0.3 s per case)"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
os.environ["RM_SPECIALIZE"] = "off"
import helpers as H

dev = "cuda"
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200


def same(a, b):
    return a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.contiguous().view(torch.int32), b.contiguous().view(torch.int32))


bad, frames = [], 0
for seed in range(n_seeds):
    gen = torch.Generator().manual_seed(770000 + seed)
    spec = H.random_spec(gen)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    n = r(1, 2)
    h, w = r(260, 560), r(260, 600)                 # >= 4096 tiles for most: the dealing orders are kept and used
    steps = 4 * r(2, 24)
    per_ray = bool(r(0, 1))
    tile = H.make_loop(H.spec_to_module(spec), h, w, n=n, regen=False, adaptive_order=0)
    pool = H.make_loop(H.spec_to_module(spec), h, w, n=n, regen=True, order_per_ray=per_ray, adaptive_order=2)
    for pose in range(2):
        q = torch.nn.functional.normalize(torch.randn(n, 4, generator=gen), dim=-1).to(dev)
        t = ((torch.rand(n, 3, generator=gen) * 2 - 1) * (4.0 if pose == 0 else 0.7)).to(dev)
        mode = [0, 1, 2, 4, 5, 6, 7][r(0, 6)]
        with torch.no_grad():
            want = tile(q, t, mode, 2, steps)
            for frame in range(3):
                frames += 1
                if not same(pool(q, t, mode, 2, steps), want):
                    bad.append((seed, pose, mode, frame, n, h, w, steps, per_ray))
    del tile, pool
    if seed % 20 == 19:
        print(f"{seed + 1} trees, {frames} frames compared, mismatches so far: {len(bad)}", flush=True)
print(f"{n_seeds} random trees, {frames} pool-kernel frames compared with the tile kernel: {len(bad)} mismatches {bad[:5]}")
sys.exit(1 if bad else 0)
