"""Where does the recording forward of BASELINE config 4 (closed make_test_scene, 512x512x64) spend its
time?  Calls rm_render_forward directly: trajectory on/off x early-out on/off x RM_CULL, HIP-event timed."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ray_marching_amd import _abi, ops, specialize
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.compiler import compiled_for
from ray_marching_amd.scene.scene_registry import make_closed_test_scene
PX, W, H, STEPS = 3.45e-6, 512, 512, 64
dev = torch.device("cuda:0")
os.environ["RM_SPECIALIZE"] = "jit"
for cull in ("0", "1"):
    os.environ["RM_CULL"] = cull
    loop = RenderLoop(make_closed_test_scene(), num_cameras=1, px_width=W, px_height=H, focal_length=PX*H,
                      sensor_width=PX*W, sensor_height=PX*H, normals_eps=5e-2).to(dev)
    cs = compiled_for(loop.scene)
    prm = cs.pack_params(dev)
    q = torch.tensor([[1.0, 0, 0, 0]], device=dev); t = torch.tensor([[0.0, 0.0, -1.0]], device=dev)
    image = torch.empty(1, H, W, 3, device=dev); nexec = torch.empty(H*W, dtype=torch.int32, device=dev)
    pfin = torch.empty(1, H, W, 3, device=dev); traj = torch.empty(STEPS, H*W, 3, device=dev)
    mm = torch.empty(_abi.WORK_WORDS, dtype=torch.int32, device=dev)
    s, keep = cs.scene_struct(prm, dev)
    cam = ops.camera_struct(loop.camera.ray_positions, loop.camera.ray_directions)
    st = _abi.current_stream(dev)
    lib = cs.lib(False, "exact")
    for with_traj in (False, True):
        for early in (True, False):
            flags = ops.default_flags(early, True)
            def run():
                _abi.lib.rm_minmax_init(_abi.ptr(mm), st)
                rc = lib.rm_render_forward(s, cam, loop.normals.tetra(), _abi.ptr(q), _abi.ptr(t), _abi.ptr(image), _abi.ptr(pfin),
                                           _abi.ptr(traj) if with_traj else None, _abi.ptr(nexec), _abi.ptr(mm), None, 0, 0, 1, STEPS, 0, H, flags, st)
                assert rc == 0
            for _ in range(3): run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            print(f"cull={cull} traj={with_traj} early={early}: {e0.elapsed_time(e1)/20*1e3:.1f} us/frame  mean nexec {nexec.float().mean().item():.1f}  specialised={cs.specialised}")
