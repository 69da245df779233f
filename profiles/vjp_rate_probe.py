"""How fast is one scene VJP?  k_sdf_bwd over N points of the closed config-4 scene (a plain loop of scene.vjp per 64 points
per wave, coalesced loads) next to k_sdf_fwd, timed with HIP events: the yardstick for k_bwd_hard_b (694 k pairs in 48 us)
and k_bwd_hard_n (1.14 M point gradients in 48 us).   python profiles/vjp_rate_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.scene.scene_registry import make_closed_test_scene
from ray_marching_amd.compiler import compiled_for

dev = torch.device("cuda:0")
scene = make_closed_test_scene().to(dev)
print("specialised:", compiled_for(scene).specialised)
gen = torch.Generator().manual_seed(0)
for n in (65536, 262144, 694000, 1140000, 4194304):
    pts = (torch.rand(n, 3, generator=gen) * 4 - 2).to(dev)
    g = torch.randn(n, 1, generator=gen).to(dev)
    def fwd():
        with torch.no_grad():
            return scene(pts)
    def bwd():
        for p in scene.parameters():
            p.grad = None
        d = scene(pts)
        e0.record()
        d.backward(g)
        e1.record()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        bwd()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        bwd(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3)
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fwd(); torch.cuda.synchronize()
    f0.record(); fwd(); f1.record(); torch.cuda.synchronize()
    print(f"n = {n:8d}: backward (k_sdf_bwd + reductions) {sorted(ts)[2]:8.1f} us = {n / sorted(ts)[2]:7.1f} M VJP/s; forward {f0.elapsed_time(f1) * 1e3:7.1f} us")
