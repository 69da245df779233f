"""Fuzz of the exactness of CULL_LSE (exact culling inside smooth unions): random blobs (tests/helpers.random_blob_spec:
8-40 affine-placed random subtrees under one SDFSmoothUnion, un-normalised quaternions, nested unions / smooth unions
inside the children) compiled with the culling (RM_CULL_LSE=1, from 2 children on; RM_CULL_UNION_TABLE=1: a blob under a
min-union is also skipped as a whole from its children's own bounds) and without (both 0); values at
16 k points and point gradients must be bit-identical, parameter gradients equal to summation order (the longer
program changes the interpreter's LDS footprint and with it the block size, see tests/fuzz_cull.py), and frames rendered
through the interpreter bit-identical.
    python tests/fuzz_cull_lse.py [n_seeds]      (a script, not collected by pytest)"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import torch
os.environ["RM_SPECIALIZE"] = "off"
os.environ["RM_CULL_LSE_MIN"] = "2"
import helpers as H
from oracle import sdf_oracle as O
from ray_marching_amd import _abi
from ray_marching_amd.compiler import compiled_for

dev = torch.device("cuda:0")
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
same = lambda x, y: torch.equal(torch.nan_to_num(x, nan=1234.5), torch.nan_to_num(y, nan=1234.5)) and torch.equal(x.isnan(), y.isnan())
bad, sites, union_sites = [], 0, 0
for seed in range(n_seeds):
    gen = torch.Generator().manual_seed(700000 + seed)
    spec, spread = H.random_blob_spec(gen)
    spec = O.map_spec(spec, lambda x: x.clone().float())
    pts = torch.cat([(torch.rand(8192, 3, generator=gen) * 2 - 1) * 2.5 * spread,
                     (torch.rand(8192, 3, generator=gen) * 2 - 1) * 0.5 * spread]).to(dev)
    # coherent waves (64 consecutive points close together), as the rays of a tile are: culls need the whole wave to agree
    pts = (pts.view(-1, 64, 3)[:, :1] + 0.02 * spread * torch.randn(pts.shape[0] // 64, 64, 3, generator=gen).to(dev)).reshape(-1, 3)
    wts = torch.randn(pts.shape[0], 1, generator=gen).to(dev)
    res = {}
    for cull in ("0", "1"):
        os.environ["RM_CULL_LSE"] = cull
        os.environ["RM_CULL_UNION_TABLE"] = cull         # ... and the whole-union test from the children's own bounds
        module = H.spec_to_module(spec).to(dev)
        cs = compiled_for(module)
        rows_ = cs.program.reshape(-1, 4)
        n_sites = int((rows_[:, 0] == _abi.OP_CULL_LSE).sum())
        if cull == "1":
            union_sites += int(((rows_[:, 0] == _abi.OP_CULL_MIN) & (rows_[:, 1] == 1)).sum())
        p = pts.clone().requires_grad_(True)
        try:
            d = module(p)
            (d * wts).sum().backward()
        except _abi.RmError as e:          # a tree too large for the interpreter's backward (accumulators in LDS): values only
            if "LDS" not in str(e):
                raise
            with torch.no_grad():
                d = module(pts)
            p.grad = torch.zeros_like(pts)
        loop = H.make_loop(module, 40, 56)
        q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev); t = torch.tensor([[0.0, 0.0, -2.0 * spread]], device=dev)
        with torch.no_grad():
            frames = [loop(q, t, m, 1, 64) for m in (4, 0, 2)]
        res[cull] = (d.detach(), p.grad, [None if x.grad is None else x.grad.clone() for x in module.parameters()], n_sites, frames)
    a, b = res["0"], res["1"]
    assert a[3] == 0 and b[3] > 0
    sites += b[3]
    gmax = max([float(torch.nan_to_num(x).abs().max()) for x in a[2] if x is not None] + [0.0])
    def close(x, y):
        fx, fy = torch.nan_to_num(x.double()), torch.nan_to_num(y.double())
        return torch.equal(x.isnan(), y.isnan()) and float((fx - fy).abs().max()) <= 1e-3 * float(fx.abs().max()) + 2e-6 * gmax + 1e-12
    ok = same(a[0], b[0]) and same(a[1], b[1]) and all(same(x, y) for x, y in zip(a[4], b[4])) \
        and all((x is None) == (y is None) and (x is None or close(x, y)) for x, y in zip(a[2], b[2]))
    if not ok:
        bad.append(seed)
        print(f"seed {seed}: MISMATCH ({b[3]} sites) values {same(a[0], b[0])} point grads {same(a[1], b[1])} "
              f"frames {[same(x, y) for x, y in zip(a[4], b[4])]}", flush=True)
print(f"{n_seeds} random blobs, {sites} CULL_LSE sites, {union_sites} smooth unions culled as a whole from their children's bounds: {len(bad)} mismatches {bad}")
sys.exit(1 if bad else 0)
