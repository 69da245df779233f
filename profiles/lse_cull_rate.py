"""How often could exact culling fire in the config-5 scene?  CPU emulation on the oracle (no GPU): 312 8x8 tiles of
the 7680x4320 frame, camera (0,0,-4.5), marched with the oracle; at every step of every tile still moving, the wave-level
tests of rm_device.h are evaluated from the children's bounding spheres:
  (a) CULL_LSE: child i of the smooth union skipped when k (lb_i - min_j ub_j) > 105 (its logsumexp term is exactly +0);
  (b) whole smooth union skipped in the outer min-union when  min_i lb_i - log(n)/k  >= room distance for all 64 rays
      (the per-child form of CULL_MIN's bound, instead of ONE sphere around all 32 children).
    python profiles/lse_cull_rate.py > profiles/r03_lse_cull_rate.txt"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle import sdf_oracle as O  # noqa: E402

torch.set_num_threads(8)
spec = O.scene_many(32)
room, blob = spec[2][0], spec[2][1]
k = float(blob[1]["blend_k"]); kids = blob[2]; n = len(kids)


def sphere(c):
    """bounding sphere (centre, radius) of an affine-wrapped primitive in the union's frame"""
    t, q = c[1]["translation"], c[1]["orientation"]
    kind, prm = c[2][0], c[2][1]
    if kind == "sphere":
        return t, float(prm["radius"])
    if kind == "box":
        return t, float(prm["halfsides"].norm())
    if kind == "torus":
        return t, float(prm["radius1"] + prm["radius2"])
    a, b = prm["start"], prm["end"]
    mid = O.quat_rotate((a + b) / 2, q) + t          # affine: child(rot(p - t, conj(q))) -> centre rot(mid, q) + t
    return mid, float((b - a).norm() * 0.5 + prm["radius"])


cent = torch.stack([sphere(c)[0] for c in kids]); R = torch.tensor([sphere(c)[1] for c in kids])
# ONE sphere around everything (what CULL_MIN uses today for the smooth union as a whole)
c_all = cent.mean(0); R_all = float(((cent - c_all).norm(dim=-1) + R).max()) + math.log(n) / k
PX = 3.45e-6; W, H, S = 7680, 4320, 256
tys = torch.arange(20, H // 8, 40); txs = torch.arange(20, W // 8, 40)
ty, tx = (a.flatten() for a in torch.meshgrid(tys, txs, indexing="ij"))
lane = torch.arange(64)
rows = (ty[:, None] * 8 + lane[None] // 8).float(); cols = (tx[:, None] * 8 + lane[None] % 8).float()
org = torch.stack([((2 * cols + 1) / W - 1) * PX * W / 2, -((2 * rows + 1) / H - 1) * PX * H / 2, torch.zeros_like(cols)], -1)
d = torch.nn.functional.normalize(org - torch.tensor([0, 0, -PX * H]), dim=-1)
p = org + torch.tensor([0.0, 0.0, -4.5])
evals = lse_children = lse_culled = union_culled = union_culled_one_sphere = 0
with torch.no_grad():
    for step in range(S):
        f = O.sdf_eval(spec, p)
        moving = (f.abs().squeeze(-1) > 1e-6).any(dim=1)          # proxy for the bit-exact early-out of the tile
        if not moving.any():
            break
        c0 = p[:, 0]; rho = (p - c0[:, None]).norm(dim=-1).max(dim=1).values
        t = (c0[:, None, :] - cent[None]).norm(dim=-1)
        lb = (t - rho[:, None]).clamp(min=0) - R[None]; ub = t + rho[:, None] + R[None]
        cm = k * (lb - ub.min(dim=1).values[:, None]) > 105
        d_room = O.sdf_eval(room, p).squeeze(-1)                  # running minimum when the smooth union is reached
        whole = (lb.min(dim=1).values - math.log(n) / k)[:, None] >= d_room
        one = ((c0 - c_all).norm(dim=-1) - rho - R_all)[:, None] >= d_room
        evals += int(moving.sum()); lse_children += int(moving.sum()) * n; lse_culled += int(cm[moving].sum())
        union_culled += int(whole.all(dim=1)[moving].sum()); union_culled_one_sphere += int(one.all(dim=1)[moving].sum())
        p = f * d + p
print(f"config-5 scene: {n} children, blend_k {k:g}, 104/k = {104 / k:.2f} units; {len(ty)} tiles, {evals} (tile, step) evaluations until every tile settled")
print(f"(a) CULL_LSE: {lse_culled} of {lse_children} child evaluations skippable = {lse_culled / lse_children:.1%}")
print(f"(b) whole smooth union skippable from the per-child bounds: {union_culled} of {evals} evaluations = {union_culled / evals:.1%}"
      f"   (with ONE sphere around all children, today's CULL_MIN: {union_culled_one_sphere / evals:.1%})")
