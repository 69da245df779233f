"""Shared test helpers: oracle spec <-> product module conversion, fixtures, tolerances."""
import os

import numpy as np
import torch

from oracle import sdf_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PX = 3.45e-6
EPS = 5e-2


def gold(name):
    return np.load(os.path.join(GOLD, name))


def spec_to_module(spec):
    """Oracle scene spec -> ray_marching_amd nn.Module tree (parameter values copied)."""
    from ray_marching_amd.scene import primitives as P, transformations as T
    kind, prm = spec[0], spec[1]

    def f(x):
        return x.detach().tolist()

    if kind == "sphere":
        return P.SDFSphere(f(prm["radius"]))
    if kind == "box":
        return P.SDFBox(tuple(f(prm["halfsides"])))
    if kind == "plane":
        return P.SDFPlane()
    if kind == "line":
        return P.SDFLine(tuple(f(prm["start"])), tuple(f(prm["end"])), f(prm["radius"]))
    if kind == "disk":
        return P.SDFDisk(f(prm["radius"]))
    if kind == "torus":
        return P.SDFTorus(f(prm["radius1"]), f(prm["radius2"]))
    if kind == "affine":
        return T.SDFAffineTransformation(spec_to_module(spec[2]), orientation=f(prm["orientation"]),
                                         translation=f(prm["translation"]))
    if kind == "smooth_union":
        return T.SDFSmoothUnion([spec_to_module(c) for c in spec[2]], f(prm["blend_k"]))
    if kind == "union":
        return T.SDFUnion([spec_to_module(c) for c in spec[2]])
    if kind == "rounding":
        return T.SDFRounding(spec_to_module(spec[2]), f(prm["rounding"]))
    if kind == "onion":
        return T.SDFOnion(spec_to_module(spec[2]), f(prm["radius"]))
    raise ValueError(kind)


def node_specs():
    from oracle.gen_golden import node_specs as ns
    return ns()


def make_loop(module, h, w, n=1, device="cuda", **kw):
    from ray_marching_amd.control import RenderLoop
    return RenderLoop(module, num_cameras=n, px_width=w, px_height=h, focal_length=PX * h,
                      sensor_width=PX * w, sensor_height=PX * h, normals_eps=EPS, **kw).to(device)


def report(name, got, want):
    """max abs error, fraction of elements beyond 1e-5 (NaN positions must agree)."""
    got = torch.as_tensor(got).detach().double().cpu()
    want = torch.as_tensor(want).detach().double().cpu()
    assert got.shape == want.shape, (name, got.shape, want.shape)
    nan_g, nan_w = torch.isnan(got), torch.isnan(want)
    assert torch.equal(nan_g, nan_w), f"{name}: NaN pattern differs ({nan_g.sum()} vs {nan_w.sum()})"
    err = (torch.nan_to_num(got) - torch.nan_to_num(want)).abs()
    err = torch.where(torch.isinf(got) & (got == want), torch.zeros_like(err), err)
    return float(err.max()) if err.numel() else 0.0, float((err > 1e-5).double().mean()) if err.numel() else 0.0


def ulp_distance(got, want):
    """|got - want| in fp32 units in the last place (bit patterns as ordered integers), int64 tensor."""
    def ordered(t):
        b = torch.as_tensor(t).detach().float().cpu().contiguous().view(torch.int32).long()
        return torch.where(b < 0, -(b & 0x7fffffff), b)
    return (ordered(got) - ordered(want)).abs()


def random_spec(gen, depth=0, max_depth=4):
    """Random SDF tree over all 11 node types (oracle spec); `gen` is a torch.Generator."""
    def u(lo, hi, n=None):
        x = torch.rand(n or 1, generator=gen) * (hi - lo) + lo
        return x if n else x[0]

    def leaf():
        k = int(torch.randint(0, 6, (1,), generator=gen))
        if k == 0:
            return ("sphere", {"radius": u(0.2, 1.0).clone()})
        if k == 1:
            return ("box", {"halfsides": u(0.1, 0.9, 3)})
        if k == 2:
            return ("plane", {})
        if k == 3:
            return ("line", {"start": u(-1.0, 1.0, 3), "end": u(-1.0, 1.0, 3) + 0.5, "radius": u(0.05, 0.3).clone()})
        if k == 4:
            return ("disk", {"radius": u(0.3, 1.0).clone()})
        return ("torus", {"radius1": u(0.5, 1.2).clone(), "radius2": u(0.05, 0.3).clone()})

    if depth >= max_depth or float(torch.rand(1, generator=gen)) < 0.25:
        return leaf()
    k = int(torch.randint(0, 5, (1,), generator=gen))
    if k == 0:
        q = torch.nn.functional.normalize(torch.randn(4, generator=gen), dim=0) * u(0.9, 1.1)   # not normalised on purpose
        return ("affine", {"translation": u(-1.0, 1.0, 3), "orientation": q}, random_spec(gen, depth + 1, max_depth))
    if k == 1:
        n = int(torch.randint(1, 5, (1,), generator=gen))
        return ("smooth_union", {"blend_k": u(4.0, 30.0).clone()}, [random_spec(gen, depth + 1, max_depth) for _ in range(n)])
    if k == 2:
        n = int(torch.randint(1, 5, (1,), generator=gen))
        return ("union", {}, [random_spec(gen, depth + 1, max_depth) for _ in range(n)])
    if k == 3:
        return ("rounding", {"rounding": u(0.0, 0.2).clone()}, random_spec(gen, depth + 1, max_depth))
    return ("onion", {"radius": u(0.02, 0.2).clone()}, random_spec(gen, depth + 1, max_depth))


def spec_has(spec, kind):
    if spec[0] == kind:
        return True
    kids = spec[2] if len(spec) > 2 else []
    kids = kids if isinstance(kids, list) else [kids]
    return any(spec_has(c, kind) for c in kids)


def random_blob_spec(gen, n_min=8, n_max=40):
    """A smooth union of many affine-placed random subtrees spread widely enough (relative to 104 / blend_k) that the
    exact logsumexp culling (RM_OP_CULL_LSE) finds children to skip, optionally inside a room and next to other
    nodes.  Oracle spec; `gen` is a torch.Generator."""
    def u(lo, hi, n=None):
        x = torch.rand(n or 1, generator=gen) * (hi - lo) + lo
        return x if n else x[0]

    k = float(u(3.0, 60.0))
    n = int(torch.randint(n_min, n_max + 1, (1,), generator=gen))
    spread = float(u(0.4, 3.0)) * 104.0 / k
    kids = []
    for _ in range(n):
        q = torch.nn.functional.normalize(torch.randn(4, generator=gen), dim=0) * u(0.93, 1.07)
        child = random_spec(gen, depth=2, max_depth=4)
        kids.append(("affine", {"translation": u(-spread, spread, 3), "orientation": q}, child))
    blob = ("smooth_union", {"blend_k": torch.tensor(k)}, kids)
    form = int(torch.randint(0, 3, (1,), generator=gen))
    if form == 0:
        return blob, spread
    if form == 1:
        room = ("onion", {"radius": torch.tensor(0.1)}, ("box", {"halfsides": torch.full((3,), 2.5 * spread + 3.0)}))
        return ("union", {}, [room, blob]), spread
    q = torch.nn.functional.normalize(torch.randn(4, generator=gen), dim=0)
    return ("union", {}, [("affine", {"translation": u(-1.0, 1.0, 3), "orientation": q}, blob), random_spec(gen, depth=2)]), spread
