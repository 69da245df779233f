"""Per-ray step counts (what k_march_regen records for its dealing order) as grey images, camera (0,0,1) and (0,0,-3):
gpurun_out/costmap_z*.png, plus how much of the 'tiles with long rays' set survives a one-pixel camera move."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_test_scene2
dev = torch.device("cuda:0")
h, w = 1080, 1920
q = torch.tensor([[1.0, 0, 0, 0]], device=dev)


def costs(t):
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=bench.PX * h, sensor_width=bench.PX * w,
                      sensor_height=bench.PX * h, normals_eps=bench.EPS, regen=True).to(dev)
    with torch.no_grad():
        loop(q, t, 4, 1, 128)
    torch.cuda.synchronize()
    (st,) = [s for s in loop._order_state.values() if s.get("T")]
    c = st["cost"].cpu().view(h // 8, w // 8, 8, 8).permute(0, 2, 1, 3).reshape(h, w)      # slot = tile * 64 + lane of the 8x8 tile
    return c


def write_pgm(path, img):
    with open(path, "wb") as f:
        f.write(b"P5 %d %d 255\n" % (img.shape[1], img.shape[0]))
        f.write(img.astype(np.uint8).tobytes())


for z in (1.0, -3.0):
    a = costs(torch.tensor([[0.0, 0.0, z]], device=dev))
    b = costs(torch.tensor([[0.003, 0.0, z]], device=dev))
    img = (a.float() / 128 * 255).numpy()
    write_pgm(f"gpurun_out/costmap_z{z:+g}.pgm", img[::2, ::2])
    la, lb = (a >= 96), (b >= 96)
    print(f"z={z:+g}: long rays {int(la.sum())} / {int(lb.sum())} after a move of 0.003; the same pixels long in both: {int((la & lb).sum())}")
    for ts in (8, 16, 32, 64):
        ta = la.view(h // ts if h % ts == 0 else -1, ts, w // ts, ts).any(dim=3).any(dim=1) if h % ts == 0 else None
        if ta is None:
            hh = (h // ts) * ts
            ta = la[:hh].view(hh // ts, ts, w // ts, ts).any(dim=3).any(dim=1)
            tb = lb[:hh].view(hh // ts, ts, w // ts, ts).any(dim=3).any(dim=1)
            na = la[:hh].view(hh // ts, ts, w // ts, ts).sum(dim=(1, 3)).float()
            nb = lb[:hh].view(hh // ts, ts, w // ts, ts).sum(dim=(1, 3)).float()
        else:
            tb = lb.view(h // ts, ts, w // ts, ts).any(dim=3).any(dim=1)
            na = la.view(h // ts, ts, w // ts, ts).sum(dim=(1, 3)).float()
            nb = lb.view(h // ts, ts, w // ts, ts).sum(dim=(1, 3)).float()
        corr = float(torch.corrcoef(torch.stack([na.flatten(), nb.flatten()]))[0, 1])
        print(f"   {ts}x{ts} tiles with a long ray: {int(ta.sum())} / {int(tb.sum())}, in both {int((ta & tb).sum())}; correlation of the long-ray counts {corr:.3f}")
