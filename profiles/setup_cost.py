"""Fixed cost of a kernel's scene set-up (parameter gather through the pointer table, derive_constants, barriers):
module(points) on 64 points = one block whose work is one evaluation, under rocprofv3 --kernel-trace.
    python profiles/setup_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd.scene.scene_registry import make_closed_test_scene, make_many_primitive_scene, make_test_scene2
dev = torch.device("cuda:0")
for name, make in (("closed1", make_closed_test_scene), ("scene2", make_test_scene2), ("many32", lambda: make_many_primitive_scene(32))):
    m = make().to(dev)
    p = torch.randn(64, 3, device=dev)
    for _ in range(5):
        pp = p.clone().requires_grad_(True)
        m(pp).sum().backward()
    torch.cuda.synchronize()
    print(name, "done")
