"""Does CULL_MIN fire on the device?  Times scene(p) (rm_sdf_forward, specialised kernels) on 16 M points
close to the far wall of the room -- where every wave should skip the object group -- with RM_CULL=0/1."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
os.environ["RM_SPECIALIZE"] = "jit"
from ray_marching_amd.scene.scene_registry import make_closed_test_scene, make_test_scene2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
n = 1 << 24
wall = torch.rand(n, 3, generator=g) * torch.tensor([8.0, 8.0, 0.2]) + torch.tensor([-4.0, -4.0, 4.6])
mid = torch.rand(n, 3, generator=g) * 2 - 1
for name, mk in (("scene2", make_test_scene2), ("scene1c", make_closed_test_scene)):
    for cull in ("0", "1"):
        os.environ["RM_CULL"] = cull
        scene = mk().to(dev)
        for label, pts in (("wall", wall), ("middle", mid)):
            p = pts.to(dev)
            with torch.no_grad():
                for _ in range(3): d = scene(p)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize(); e0.record()
                for _ in range(10): d = scene(p)
                e1.record(); torch.cuda.synchronize()
            print(f"{name} cull={cull} {label}: {e0.elapsed_time(e1)/10*1e3:.1f} us / 16M points, checksum {d.double().sum().item():.6f}")
