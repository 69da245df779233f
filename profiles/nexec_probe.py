import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ray_marching_amd import _abi, ops
from ray_marching_amd.control import RenderLoop
from ray_marching_amd.compiler import compiled_for
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H = 3.45e-6, 1920, 1080
dev = torch.device("cuda:0")
for tile8 in (True, False):
  for z in (-3.0, 1.0):
    loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=W, px_height=H, focal_length=PX*H, sensor_width=PX*W, sensor_height=PX*H, normals_eps=5e-2, tile8x8=tile8).to(dev)
    cs = compiled_for(loop.scene)
    prm = cs.pack_params(dev)
    q = torch.tensor([[1.0,0,0,0]], device=dev); t = torch.tensor([[0.0,0.0,z]], device=dev)
    image = torch.empty(1,H,W,3, device=dev); nexec = torch.empty(H*W, dtype=torch.int32, device=dev)
    mm = torch.empty(_abi.WORK_WORDS, dtype=torch.int32, device=dev)
    s, keep = cs.scene_struct(prm, dev)
    cam = ops.camera_struct(loop.camera.ray_positions, loop.camera.ray_directions)
    st = _abi.current_stream(dev)
    _abi.lib.rm_minmax_init(_abi.ptr(mm), st)
    rc = cs.lib().rm_render_forward(s, cam, loop.normals.tetra(), _abi.ptr(q), _abi.ptr(t), _abi.ptr(image), None, None, _abi.ptr(nexec), _abi.ptr(mm), None, 0, 4, 1, 128, 0, H, ops.default_flags(True, tile8), st)
    torch.cuda.synchronize()
    ne = nexec.float()
    hist = torch.histc(ne, bins=8, min=0, max=128)
    print(f"tile8x8={tile8} cam z={z}: mean steps {ne.mean().item():.1f} median {ne.median().item():.0f} max {ne.max().item():.0f} frac==128: {(ne==128).float().mean().item():.3f} hist(16-step bins)={[int(x) for x in hist.tolist()]}")
