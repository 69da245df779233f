#!/usr/bin/env python3
"""BASELINE config 4: differentiable rendering on the MI355X.

Scene = SDFUnion([make_test_scene(), room]) (closed, so every ray hits).  The target is a
render of the same scene with its 3 translations perturbed by N(0, 0.05^2) and its 3 quaternions by
N(0, 0.02^2) (torch seed 0).  Loss = MSE of the Lambertian image (mode 0); Adam on the 6 pose parameters.
Every forward is one k_render_fwd launch (with trajectory recording), every backward one k_render_bwd.

    python examples/optimize_scene.py --size 512 --steps 64 --iters 200

Note on conditioning (measured, profiles/lr_scan.py): the exact gradient of the discrete 64-step
render -- what the reference's autograd computes and what these kernels reproduce to 6e-7 -- is
dominated by a few crease / grazing pixels where the eps = 0.05 tetrahedral normal is normalised from a
tiny vector (|grad| ~ 6 while finite differences of the loss see a slope ~0.05).  Plain Adam therefore
needs a small step (1e-3) and descends slowly; that is a property of the reference's formulation.
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ray_marching_amd.control import RenderLoop  # noqa: E402
from ray_marching_amd.scene.scene_registry import make_closed_test_scene  # noqa: E402

PX, EPS = 3.45e-6, 5e-2


def pose_parameters(scene):
    return [p for n, p in scene.named_parameters() if n.endswith("translation") or n.endswith("orientation")]


def make_problem(size, device, seed=0):
    target_scene = make_closed_test_scene()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for n, p in target_scene.named_parameters():
            if n.endswith("translation"):
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            elif n.endswith("orientation"):
                p.add_(torch.randn(p.shape, generator=g) * 0.02)
    scene = make_closed_test_scene()
    kw = dict(num_cameras=1, px_width=size, px_height=size, focal_length=PX * size, sensor_width=PX * size,
              sensor_height=PX * size, normals_eps=EPS)
    return RenderLoop(scene, **kw).to(device), RenderLoop(target_scene, **kw).to(device)


def run(size=512, march_steps=64, iters=200, lr=1e-3, device="cuda", log=print):
    loop, target_loop = make_problem(size, device)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=device)
    t = torch.tensor([[0.0, 0.0, -1.0]], device=device)
    with torch.no_grad():
        target = target_loop(q, t, 0, 1, march_steps)[..., :1]
    params = pose_parameters(loop.scene)
    for p in loop.scene.parameters():
        p.requires_grad_(any(p is x for x in params))
    opt = torch.optim.Adam(params, lr=lr)
    losses, fwd_ms, bwd_ms = [], [], []
    for it in range(iters):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        opt.zero_grad(set_to_none=True)
        e0.record()
        loss = (loop(q, t, 0, 1, march_steps)[..., :1] - target).pow(2).mean()
        e1.record()
        loss.backward()
        e2.record()
        opt.step()
        torch.cuda.synchronize()
        losses.append(loss.item()); fwd_ms.append(e0.elapsed_time(e1)); bwd_ms.append(e1.elapsed_time(e2))
        if it % max(1, iters // 10) == 0:
            log(f"iter {it:4d}  loss {losses[-1]:.3e}  fwd {fwd_ms[-1]:.3f} ms  bwd {bwd_ms[-1]:.3f} ms")
    return {"loss_first": losses[0], "loss_last": losses[-1], "fwd_ms": sorted(fwd_ms)[len(fwd_ms) // 2],
            "bwd_ms": sorted(bwd_ms)[len(bwd_ms) // 2], "losses": losses}


def run_graphed(size=512, march_steps=64, iters=200, lr=1e-3, device="cuda", log=print):
    """The same optimisation with forward + backward + Adam step recorded ONCE in a HIP graph
    (torch.cuda.graph) and replayed: no Python, no launch gaps between the ~40 small kernels of a step.
    The replayed trajectory equals the eager one with Adam(capturable=True) digit for digit (checked on the
    GPU); against run() it drifts, because the two Adam implementations round differently and this
    problem is ill-conditioned (see the note at the top)."""
    loop, target_loop = make_problem(size, device)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=device)
    t = torch.tensor([[0.0, 0.0, -1.0]], device=device)
    with torch.no_grad():
        target = target_loop(q, t, 0, 1, march_steps)[..., :1]
    params = pose_parameters(loop.scene)
    for p in loop.scene.parameters():
        p.requires_grad_(any(p is x for x in params))
    opt = torch.optim.Adam(params, lr=lr, capturable=True)

    def step():
        loss = (loop(q, t, 0, 1, march_steps)[..., :1] - target).pow(2).mean()
        loss.backward()
        opt.step()
        return loss

    warm = 3
    losses = []
    from ray_marching_amd.graphs import capture_step
    # warm-up iterations are real optimiser steps; warm-up and capture share one stream (ray_marching_amd/graphs.py)
    graph, loss, _ = capture_step(lambda: losses.append(step()) or losses[-1], params, warmup=warm)
    losses[:] = [x.item() for x in losses[:warm]]
    losses.append(float("nan"))                    # the capture pass itself does not execute
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(warm + 1, iters):
        graph.replay()
        if it % max(1, iters // 10) == 0 or it == iters - 1:
            losses.append(loss.item())             # reading the loss synchronises: only every few iterations
            log(f"iter {it:4d}  loss {losses[-1]:.3e}")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / max(1, iters - warm - 1) * 1e3
    finite = [x for x in losses if x == x]
    return {"loss_first": finite[0], "loss_last": finite[-1], "step_ms": ms, "losses": finite}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--graph", action="store_true", help="record forward + backward + Adam in one HIP graph and replay it")
    a = ap.parse_args()
    t0 = time.time()
    if a.graph:
        r = run_graphed(a.size, a.steps, a.iters, a.lr)
        print(f"loss {r['loss_first']:.3e} -> best {min(r['losses']):.3e} / last {r['loss_last']:.3e} in {a.iters} iterations "
              f"({time.time() - t0:.1f} s wall); {r['step_ms']:.3f} ms per optimiser step (graph replay) at {a.size}x{a.size}x{a.steps}")
        sys.exit(0)
    r = run(a.size, a.steps, a.iters, a.lr)
    print(f"loss {r['loss_first']:.3e} -> best {min(r['losses']):.3e} / last {r['loss_last']:.3e} in {a.iters} iterations "
          f"({time.time() - t0:.1f} s wall); median fwd {r['fwd_ms']:.3f} ms, bwd {r['bwd_ms']:.3f} ms at {a.size}x{a.size}x{a.steps}")
