"""Where does the S=0 fixed cost of k_render_fwd come from?  (rocprof-free: many reps, GPU-bound check)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd import _abi, ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.compiler import compiled_for
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H = 3.45e-6, 1920, 1080
dev = torch.device("cuda:0")
loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX*H, sensor_width=PX*W, sensor_height=PX*H, normals_eps=5e-2).to(dev)
cs = compiled_for(loop.scene); prm = cs.pack_params(dev)
q = torch.tensor([[1.0,0,0,0]], device=dev); t = torch.tensor([[0.0,0.0,-3.0]], device=dev)
image = torch.empty(1,H,W,3, device=dev); mm = torch.empty(_abi.WORK_WORDS, dtype=torch.int32, device=dev)
s, keep = cs.scene_struct(prm, dev)
cam = ops.camera_struct(loop.camera.ray_positions, loop.camera.ray_directions)
st = _abi.current_stream(dev)
tet = loop.normals.tetra()
def run(flags, mode, S, use_work=True, reps=20):
    def once():
        if use_work: _abi.lib.rm_minmax_init(_abi.ptr(mm), st)
        rc = cs.lib().rm_render_forward(s, cam, tet, _abi.ptr(q), _abi.ptr(t), _abi.ptr(image), None, None, None, _abi.ptr(mm) if use_work else None, None, 0, mode, 1, S, 0, H, flags, st)
        assert rc == 0
    for _ in range(3): once()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): once()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, flags in (("linear static", 1), ("8x8 static", 3), ("linear dynamic", 5), ("8x8 dynamic", 7)):
    print(f"{name:16s} mode3 S=0: {run(flags,3,0):6.1f} us   mode4 S=0: {run(flags,4,0):6.1f} us   mode4 S=128: {run(flags,4,128):6.1f} us")
