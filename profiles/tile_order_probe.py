"""Longest-first tile order from the previous frame's per-tile step counts: how much of the straggler tail
does it remove?  (config 2; needs a library built with the rm_debug_tile_order hook)"""
import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ray_marching_amd import _abi, ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.compiler import compiled_for
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H, STEPS = 3.45e-6, 1920, 1080, 128
dev = torch.device("cuda:0")
for z in (-3.0, 1.0):
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX*H, sensor_width=PX*W,
                      sensor_height=PX*H, normals_eps=5e-2).to(dev)
    cs = compiled_for(loop.scene); lib = cs.lib(False, "exact")
    
    prm = cs.pack_params(dev)
    q = torch.tensor([[1.0, 0, 0, 0]], device=dev); t = torch.tensor([[0.0, 0.0, z]], device=dev)
    image = torch.empty(1, H, W, 3, device=dev); nexec = torch.empty(H*W, dtype=torch.int32, device=dev)
    mm = torch.empty(_abi.WORK_WORDS, dtype=torch.int32, device=dev)
    s, keep = cs.scene_struct(prm, dev)
    cam = ops.camera_struct(loop.camera.ray_positions, loop.camera.ray_directions)
    st = _abi.current_stream(dev)
    flags = ops.default_flags(True, True, True)
    def run(want_nexec=False):
        _abi.lib.rm_minmax_init(_abi.ptr(mm), st)
        rc = lib.rm_render_forward(s, cam, loop.normals.tetra(), _abi.ptr(q), _abi.ptr(t), _abi.ptr(image), None, None,
                                   _abi.ptr(nexec) if want_nexec else None, _abi.ptr(mm), None, 0, 4, 1, STEPS, 0, H, flags, st)
        assert rc == 0
    def timeit(n=30):
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    lib.rm_debug_tile_order(None)
    run(True); torch.cuda.synchronize()
    ref = image.clone()
    base = timeit()
    cost = nexec.view(H // 8, 8, W // 8, 8)[:, 0, :, 0].reshape(-1)
    order = torch.argsort(cost, descending=True, stable=True).to(torch.int32).contiguous()
    lib.rm_debug_tile_order(_abi.ptr(order))
    run(); torch.cuda.synchronize()
    same = torch.equal(torch.nan_to_num(image), torch.nan_to_num(ref))
    lpt = timeit()
    rnd = torch.randperm(order.numel(), device=dev).to(torch.int32).contiguous()
    lib.rm_debug_tile_order(_abi.ptr(rnd)); rand = timeit()
    lib.rm_debug_tile_order(None)
    print(f"camera z={z}: natural order {base:.1f} us, longest-first {lpt:.1f} us, random {rand:.1f} us; same image: {same}; "
          f"tiles with all {STEPS} steps: {(cost == STEPS).float().mean().item():.2f}")
