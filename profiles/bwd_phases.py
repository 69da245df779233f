"""Where k_render_bwd's wave-time goes (library built with -DRM_BWD_STAMPS: s_memtime at phase boundaries, summed over
the waves into workspace words 8..15).  Config-4 step, closed scene 1, 64 steps.
    RM_SPECIALIZE=jit RM_HIPCC_EXTRA=-DRM_BWD_STAMPS RM_LIB_DIR=/tmp/rm_stamps python profiles/bwd_phases.py [size=512] [mode=0]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ray_marching_amd import ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.scene.scene_registry import make_closed_test_scene
size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda:0")
scene = make_closed_test_scene()
loop = RenderLoop(scene, num_cameras=1, px_width=size, px_height=size, focal_length=bench.PX * size,
                  sensor_width=bench.PX * size, sensor_height=bench.PX * size, normals_eps=bench.EPS).to(dev)
q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=dev); t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
target = torch.rand(1, size, size, 1, device=dev)
ops.bwd_tile_cost_sink = torch.zeros(int(ops._lib.rm_wave_tiles(1, size, size, 2)), dtype=torch.int32, device=dev)
names = ["prologue: loads + shader VJP", "normals VJP (4-5 scene VJPs with parameters)", "reverse march: first iterates + vote",
         "  anchor point gradient (vjp_point)", "  converged-tail loop (iterate windows, votes, flops)", "  parameter replay (vjp_replay)",
         "  deferral / in-place walk / frozen rest", "between set-up and the first tile", "the next tile: returning atomic on the tile queues (steals, last look)",
         "the tile's stores"]
for it in range(3):
    for p in scene.parameters(): p.grad = None
    (loop(q, t, mode, 1, 64)[..., :1] - target).pow(2).mean().backward()
    torch.cuda.synchronize()
    w = ops.bwd_last_work.cpu().tolist()[8:18]
tot = sum(w)
print(f"k_render_bwd phases at {size}x{size}, mode {mode} (share of summed wave time; last of 3 steps)")
for n, x in zip(names, w):
    print(f"  {100.0 * x / max(tot, 1):5.1f} %  {n}")
print(f"  rays deferred: {ops.bwd_last_work.cpu().tolist()[32]}")
