import sys, os, importlib.util
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
spec_ = importlib.util.spec_from_file_location("ex", os.path.join(sys.path[0], "examples", "optimize_scene.py"))
ex = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(ex)
for lr in (2e-3, 1e-3, 5e-4, 2e-4):
    r = ex.run(size=256, march_steps=64, iters=200, lr=lr, log=lambda *_: None)
    L = r["losses"]
    print(f"lr {lr:g}: {L[0]:.4e} -> min {min(L):.4e} last {L[-1]:.4e}  at 50:{L[50]:.3e} 100:{L[100]:.3e}")
# finite-difference check of the directional derivative along -grad
loop, tl = ex.make_problem(256, "cuda")
q = torch.tensor([[1.0,0,0,0]], device="cuda"); t = torch.tensor([[0.0,0.0,-1.0]], device="cuda")
with torch.no_grad(): target = tl(q,t,0,1,64)[...,:1]
params = ex.pose_parameters(loop.scene)
def loss_fn():
    return (loop(q,t,0,1,64)[...,:1]-target).pow(2).mean()
loss = loss_fn(); loss.backward()
g = [p.grad.clone() for p in params]
gn = sum((x*x).sum() for x in g).sqrt().item()
for h in (1e-2, 3e-3, 1e-3, 3e-4):
    with torch.no_grad():
        for p,x in zip(params,g): p.sub_(h*x/gn)
        l2 = loss_fn().item()
        for p,x in zip(params,g): p.add_(h*x/gn)
    print(f"step {h:g} along -grad/|grad|: dL = {l2-loss.item():+.3e}  predicted {-h*gn:+.3e}")
