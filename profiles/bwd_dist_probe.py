"""Config 4 (closed make_test_scene, 512x512x64): how far do the recording forward and the reverse sweep walk per
wave tile, and how long is each ray's converged tail (steps at which the stored iterate no longer changes)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd import _abi, ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_closed_test_scene
dev = torch.device("cuda:0")
h = w = 512; steps = 64
scene = make_closed_test_scene()
loop = RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                  sensor_height=bench.PX * h, normals_eps=bench.EPS).to(dev)
q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev); t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
target = torch.rand(1, h, w, 1, device=dev)
T = int(_abi.lib.rm_wave_tiles(1, h, w, _abi.FLAG_TILE8X8))
ops.bwd_tile_cost_sink = torch.zeros(T, dtype=torch.int32, device=dev)
img = loop(q, t, 0, 1, steps)
saved = img.grad_fn.saved_tensors            # prm, q, t, rp, rd, p_final, traj, nexec
traj, nexec = saved[6], saved[7]
(img[..., :1] - target).pow(2).mean().backward()
torch.cuda.synchronize()
walked = ops.bwd_tile_cost_sink.float()
Hn = int(ops.bwd_last_work[32])
print(f"rays handed to the deferred-ray kernels: {Hn} of {h * w}")
hard, cap = ops.bwd_last_hard
top = hard[cap:2 * cap].view(torch.int32)[:Hn].long()
gbuf = hard[10 * cap + steps * cap * 4:10 * cap + steps * cap * 5].view(steps, cap)[:, :Hn]
need = (torch.arange(steps, device=dev)[:, None] <= top[None, :])
nz = (gbuf != 0) & need
grp = (Hn + 63) // 64
pad = grp * 64 - Hn
needp = torch.nn.functional.pad(need, (0, pad)).view(steps, grp, 64)
nzp = torch.nn.functional.pad(nz, (0, pad)).view(steps, grp, 64)
print(f"deferred (ray, step) pairs: {int(need.sum())} (mean remaining steps {top.float().mean() + 1:.1f}); with g != 0: {int(nz.sum())} "
      f"({nz.sum() / need.sum():.2f}); wave items k_bwd_hard_n: {int(needp.any(-1).sum())}, k_bwd_hard_b: {int(nzp.any(-1).sum())} "
      f"(dense packing would need {int(need.sum()) // 64} / {int(nz.sum()) // 64})")
ne = nexec.view(h // 8, 8, w // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)[:, 0].float()     # per tile (wave-uniform)
print(f"forward: executed steps per tile  mean {ne.mean():.1f}  p50 {ne.median():.0f}  p90 {ne.quantile(.9):.0f}  max {ne.max():.0f}; tiles at {steps}: {(ne == steps).float().mean():.3f}")
print(f"reverse: walked steps per tile    mean {walked.mean():.1f}  p50 {walked.median():.0f}  p90 {walked.quantile(.9):.0f}  max {walked.max():.0f}; tiles at {steps}: {(walked == steps).float().mean():.3f}")
# per ray: first step from which the iterate is bitwise constant (within the recorded part)
same = (traj[1:] == traj[:-1]).all(-1)                       # [S-1, R]  p_{i+1} == p_i
conv = steps - 1 - same.flip(0).int().cumprod(0).sum(0)     # first i with p_j const for j >= i
print(f"per ray: first bitwise-constant step  mean {conv.float().mean():.1f}  p50 {conv.float().median():.0f}  p90 {conv.float().quantile(.9):.0f}  rays never constant: {(conv >= steps - 1).float().mean():.3f}")
hist = torch.bincount(walked.long(), minlength=steps + 1)
print("reverse walked-steps histogram (tiles):", {i: int(c) for i, c in enumerate(hist.tolist()) if c})
# classify the rays that never become constant: in a short cycle at the end of the march, or sliding?
never = conv >= steps - 1
eq = lambda a, b: (traj[a] == traj[b]).all(-1)
p2 = never & eq(63, 61) & eq(62, 60)
p3 = never & ~p2 & eq(63, 60) & eq(62, 59)
p4 = never & ~p2 & ~p3 & eq(63, 59) & eq(62, 58)
p6 = never & ~p2 & ~p3 & ~p4 & eq(63, 57)
print(f"never-constant rays: {int(never.sum())}; period 2: {int(p2.sum())}, 3: {int(p3.sum())}, 4: {int(p4.sum())}, 6: {int(p6.sum())}, "
      f"none of these (sliding / long cycle): {int((never & ~p2 & ~p3 & ~p4 & ~p6).sum())}")
# first step from which a period-2 ray is in its cycle
same2 = (traj[2:] == traj[:-2]).all(-1)                     # p_{i+2} == p_i
conv2 = steps - 2 - same2.flip(0).int().cumprod(0).sum(0)
tile_of = lambda x: x.view(h // 8, 8, w // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
c12 = torch.minimum(conv, conv2)
print(f"per tile: step from which EVERY lane is in a cycle of length 1 or 2: mean {tile_of(c12).max(1).values.float().mean():.1f}, "
      f"tiles where that never happens: {(tile_of(c12).max(1).values >= steps - 2).float().mean():.3f}")
# the tiles the reverse sweep still walks step by step: how far do their rays move per step near the end,
# and how many lanes of the tile are still moving?
hard = (walked >= 32).nonzero().flatten()
mv = (traj[1:] - traj[:-1]).abs().amax(-1)                  # [S-1, R] per-step movement
mv_t = mv.t().reshape(h // 8, 8, w // 8, 8, steps - 1).permute(0, 2, 1, 3, 4).reshape(-1, 64, steps - 1)[hard]   # [hard tiles, 64 lanes, S-1]
last = mv_t[:, :, -8:].amax(-1)                              # largest move in the last 8 steps, per lane
if hard.numel() == 0:
    print('no tile walks 32 or more steps in place any more (round 1: 15 % of the tiles walked all 64)')
for thr in (() if hard.numel() == 0 else (1e-6, 1e-5, 1e-4, 1e-3, 1e-2)):
    print(f"hard tiles ({len(hard)}): lanes still moving > {thr:g} in the last 8 steps: mean {(last > thr).float().sum(1).mean():.1f} of 64; tiles with none: {((last > thr).sum(1) == 0).float().mean():.3f}")
