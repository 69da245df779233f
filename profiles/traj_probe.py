"""Which rays have not settled after 128 steps, and what are they doing?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_marching_amd import _abi, ops
from ray_marching_amd.compiler import compiled_for
from ray_marching_amd.rendering.ray_marching import PinholeCamera
from ray_marching_amd.scene.scene_registry import make_test_scene2
PX, W, H, S = 3.45e-6, 480, 270, 128
dev = torch.device("cuda:0")
scene = make_test_scene2().to(dev)
cam = PinholeCamera(1, W, H, PX*H*4, PX*W*4, PX*H*4).to(dev)   # same FOV as 1920x1080
q = torch.tensor([[1.0,0,0,0]], device=dev); t = torch.tensor([[0.0,0.0,-3.0]], device=dev)
pos, _, _, dirs = cam(q, t)
cs = compiled_for(scene); prm = cs.pack_params(dev)
n = H*W
p = pos.reshape(-1,3).contiguous(); v = dirs.reshape(-1,3).contiguous()
out = torch.empty_like(p); traj = torch.empty(S, n, 3, device=dev)
s, keep = cs.scene_struct(prm, dev)
cs.lib().rm_march_forward(s, _abi.ptr(p), _abi.ptr(v), _abi.ptr(out), _abi.ptr(traj), None, n, S, 0, _abi.current_stream(dev))
torch.cuda.synchronize()
full = torch.cat([traj, out[None]], 0)                 # [S+1, n, 3]
bits = full.view(torch.int32)
# first step at which the state equals ANY earlier state (true cycle entry), brute force over lags 1..16
settled = torch.full((n,), S+1, dtype=torch.int64, device=dev)
for lag in range(1, 33):
    same = (bits[lag:] == bits[:-lag]).all(-1)          # [S+1-lag, n]
    first = torch.where(same.any(0), same.float().argmax(0) + lag, torch.full((n,), S+1, device=dev))
    settled = torch.minimum(settled, first)
print("rays never repeating a state within 128 steps (lags<=32):", (settled > S).float().mean().item())
print("settle-step histogram (16-bins):", torch.histc(settled.clamp(max=S).float(), bins=8, min=0, max=128).int().tolist())
bad = (settled > S)
d = scene(out).reshape(-1)
step_last = (full[-1] - full[-2]).norm(dim=-1)
print("unsettled: |f(p_S)| median", d[bad].abs().median().item(), "max", d[bad].abs().max().item(), " last step size median", step_last[bad].median().item())
print("settled:   |f(p_S)| median", d[~bad].abs().median().item())
# where are they? coarse map
m = bad.view(H, W).float()
coarse = torch.nn.functional.avg_pool2d(m[None,None], (27, 30))[0,0]
for row in coarse: print("".join(" .:-=+*#%@"[min(9,int(x*9.99))] for x in row.tolist()))
# sample trajectories of a few unsettled rays: last 6 |f| values
idx = bad.nonzero().flatten()[:: max(1, bad.sum().item()//5)][:5]
for i in idx.tolist():
    fs = [scene(full[k, i][None]).item() for k in (8, 16, 32, 64, 96, 120, 127, 128)]
    print("ray", i, "pixel", divmod(i, W), "p_S", [round(x,4) for x in out[i].tolist()], "f at steps 8,16,32,64,96,120,127,128:", ["%.2e" % x for x in fs])
