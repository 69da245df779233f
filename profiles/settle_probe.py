"""Config 2 at the reference's default pose (0,0,1) and at (0,0,-3): per 8x8 wave tile, how many of the 64 rays are
still NOT in a bitwise cycle (period <= 4) when the tile's wave would leave, and how the per-ray settle steps
spread inside a tile.  Decides whether compacting unsettled rays across waves can pay."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd import _abi, ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.compiler import compiled_for
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H, S = 3.45e-6, 1920, 1080, 128
dev = torch.device("cuda:0")
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX * H, sensor_width=PX * W,
                  sensor_height=PX * H, normals_eps=5e-2).to(dev)
cs = compiled_for(loop.scene)
q = torch.tensor([[1.0, 0, 0, 0]], device=dev)
for z in (1.0, -3.0):
    t = torch.tensor([[0.0, 0.0, z]], device=dev)
    with torch.no_grad():
        pos, _, _, dirs = loop.camera(q, t)
    n = H * W
    traj = torch.empty(S, n, 3, device=dev)
    out = torch.empty(n, 3, device=dev)
    sc, keep = cs.scene_struct(None, dev)
    rc = cs.lib().rm_march_forward(sc, _abi.ptr(pos.reshape(-1, 3).contiguous()), _abi.ptr(dirs.reshape(-1, 3).contiguous()), _abi.ptr(out),
                                   _abi.ptr(traj), None, n, S, 0, 0, _abi.current_stream(dev))     # flags 0: every step executed and stored
    assert rc == 0
    torch.cuda.synchronize()
    full = torch.cat([traj, out[None]], 0)                   # p_0 .. p_S
    settle = torch.full((n,), S, dtype=torch.int32, device=dev)
    # first step i from which p_{j+L} == p_j for all j >= i, for the smallest period L in {1, 2, 4}
    for L in (4, 2, 1):
        same = (full[L:] == full[:-L]).all(-1)               # [S+1-L, n]
        run = same.flip(0).int().cumprod(0).sum(0)           # trailing run length
        first = (S + 1 - L) - run
        settle = torch.where(run > 0, torch.minimum(settle, first.int()), settle)
    st = settle.view(H // 8, 8, W // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64).float()
    tile_exit = st.max(1).values
    never = (st >= S)
    print(f"z={z:+g}: rays never in a cycle: {never.float().mean():.4f}; tiles that run all {S} steps: {(tile_exit >= S).float().mean():.3f}; "
          f"mean tile exit step {tile_exit.mean():.1f}, mean ray settle step {st.mean():.1f}")
    hard = tile_exit >= S
    k = never[hard].float().sum(1)
    print(f"   in the {int(hard.sum())} full-length tiles: unsettled lanes per tile mean {k.mean():.1f}, median {k.median():.0f}, "
          f"p90 {k.quantile(.9):.0f}; tiles with <= 8 unsettled lanes: {(k <= 8).float().mean():.2f}, <= 16: {(k <= 16).float().mean():.2f}, <= 32: {(k <= 32).float().mean():.2f}")
    # lane-steps actually needed (each ray until its own settle step) vs executed by wave-uniform exits
    need = st.clamp(max=S).sum().item()
    done = (tile_exit.clamp(max=S) * 64).sum().item()
    print(f"   ray-steps needed {need / 1e6:.1f} M, executed with wave-uniform exits {done / 1e6:.1f} M  (x{done / need:.2f})")
    del traj, full
