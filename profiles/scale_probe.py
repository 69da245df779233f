"""How large a user scene gets through each path: make_many_primitive_scene(n) for growing n, values / point gradients /
parameter gradients against the oracle's autograd on 4096 points, through the LDS interpreter and (jit) the specialised
library hipcc builds on the spot.      python profiles/scale_probe.py [off|jit] n n n ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
os.environ["RM_SPECIALIZE"] = mode
import torch
from oracle import sdf_oracle as O
from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
from ray_marching_amd.compiler import compiled_for
from ray_marching_amd import specialize
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(7)
pts = ((torch.rand(4096, 3, generator=gen) * 2 - 1) * 4.0)
w = torch.randn(4096, 1, generator=gen)
for n in map(int, sys.argv[2:]):
    line = f"n={n:4d} [{mode}]"
    try:
        t0 = time.time()
        mod = make_many_primitive_scene(n).to(dev)
        cs = compiled_for(mod)
        if mode == "jit":
            specialize.ensure(mod)
            cs = compiled_for(mod)
        line += f" instr {cs.program.reshape(-1,4).shape[0]} params {cs.n_params} specialised={cs.specialised} build {time.time()-t0:.0f}s"
        p = pts.to(dev).requires_grad_(True)
        d = mod(p)
        ref = O.map_spec(O.scene_many(n), lambda x: x.clone().requires_grad_(True))
        with O.math_mode("restated"), torch.no_grad():
            dv = O.sdf_eval(ref, pts)
        pc = pts.clone().requires_grad_(True)
        dr = O.sdf_eval(ref, pc)
        line += f" | value maxdiff {float((d.detach().cpu() - dv).abs().max()):.1e}"
        try:
            (d * w.to(dev)).sum().backward()
            (dr * w).sum().backward()
            gp = float((p.grad.cpu() - pc.grad).abs().max())
            mine = torch.cat([x.grad.flatten().cpu() for x in mod.parameters()])
            theirs = torch.cat([x.grad.flatten() for _, x in O.spec_parameters(ref)])
            line += f" | dL/dp maxdiff {gp:.1e} | dL/dtheta maxrel {float(((mine - theirs).abs() / (1e-3 + theirs.abs())).max()):.1e} ({mine.numel()} params)"
            # the frame path: Lambertian frame 24x32, 32 steps, MSE against a fixed target, all four gradient kernels
            from ray_marching_amd.control import RenderLoop
            for x in mod.parameters(): x.grad = None
            for _, x in O.spec_parameters(ref): x.grad = None
            px = 3.45e-6
            loop = RenderLoop(mod, num_cameras=1, px_width=32, px_height=24, focal_length=px * 24, sensor_width=px * 32,
                              sensor_height=px * 24, normals_eps=5e-2).to(dev)
            q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.0, 0.0, -4.0]])
            target = torch.rand(1, 24, 32, 1, generator=gen)
            img = loop(q.to(dev), t.to(dev), 0, 1, 32)
            (img[..., :1] - target.to(dev)).pow(2).mean().backward()
            cam = O.camera_buffers(1, 32, 24, px * 24, px * 32, px * 24)
            want = O.render(ref, cam, q, t, 0, 1, 32, 5e-2)
            (want[..., :1] - target).pow(2).mean().backward()
            mine = torch.cat([x.grad.flatten().cpu() for x in mod.parameters()])
            theirs = torch.cat([x.grad.flatten() for _, x in O.spec_parameters(ref)])
            line += f" | frame maxdiff {float((img.detach().cpu() - want.detach()).abs().max()):.1e} frame dL/dtheta max abs diff {float((mine - theirs).abs().max()):.1e} of {float(theirs.abs().max()):.1e}"
            big = RenderLoop(mod, num_cameras=1, px_width=256, px_height=256, focal_length=px * 256, sensor_width=px * 256,
                             sensor_height=px * 256, normals_eps=5e-2).to(dev)
            tgt = torch.rand(1, 256, 256, 1, device=dev)
            def step():
                for x in mod.parameters(): x.grad = None
                (big(q.to(dev), t.to(dev), 0, 1, 64)[..., :1] - tgt).pow(2).mean().backward()
            step(); torch.cuda.synchronize(); t1 = time.time()
            for _ in range(3): step()
            torch.cuda.synchronize()
            line += f" | 256x256x64 fwd+bwd {(time.time() - t1) / 3 * 1e3:.1f} ms"
        except Exception as e:      # noqa: BLE001
            line += f" | BACKWARD {type(e).__name__}: {str(e)[:140]}"
    except Exception as e:          # noqa: BLE001
        line += f" | {type(e).__name__}: {str(e)[:200]}"
    print(line, flush=True)
