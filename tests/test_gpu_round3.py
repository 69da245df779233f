"""GPU tests added in round 3: main.py's torch.compile line, the config-4 shape at full size, the fused
normalisation VJP kernel, the training-step helper."""
import os
import warnings

import pytest
import torch

from oracle import sdf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _device_kernels(prof):
    return [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]


def test_main_py_torch_compile_line_is_inert():
    """/root/reference/main.py:44: ``render_loop = torch.compile(render_loop, mode='max-autotune')`` may stay.  The
    compiled wrapper returns the eager bits for every shader mode and launches exactly the eager kernels: no Inductor /
    Triton kernel, no cast or copy pass on the path."""
    from torch.profiler import ProfilerActivity, profile
    h, w, steps = 90, 160, 32
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w).to(torch.float16)       # main.py:20-26: fp16 module
    compiled = torch.compile(loop, mode="max-autotune")
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV).half()
    t = torch.tensor([[0.0, 0.0, 1.0]], device=DEV).half()
    with torch.no_grad():
        for mode in range(8):
            a = loop(q, t, mode, 1, steps)
            b = compiled(q, t, mode, 1, steps)
            assert a.dtype == b.dtype and a.shape == b.shape == (1, h, w, 3)
            assert torch.equal(torch.nan_to_num(a.float(), nan=-7.0), torch.nan_to_num(b.float(), nan=-7.0)), mode
        names = {}
        for tag, fn in (("eager", loop), ("compiled", compiled)):
            for _ in range(2):
                fn(q, t, 4, 1, steps)
            torch.cuda.synchronize()
            with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
                fn(q, t, 4, 1, steps)
                torch.cuda.synchronize()
            names[tag] = _device_kernels(prof)
    if not names["eager"]:
        pytest.skip("the profiler recorded no device activity on this box")
    print("kernels of one frame:", names)
    strip = lambda ks: sorted(k for k in ks if "k_minmax_init" not in k and "Memset" not in k)
    assert strip(names["compiled"]) == strip(names["eager"])
    assert sum("k_render_fwd" in k for k in names["compiled"]) == 1
    assert not [k for k in names["compiled"] if "triton" in k.lower() or "inductor" in k.lower()]
    # the stand-alone modules under torch.compile: same bits as eager
    scene = loop.scene
    pts = torch.rand(4096, 3, device=DEV).half() * 4 - 2
    with torch.no_grad():
        assert torch.equal(torch.compile(scene)(pts), scene(pts))


@pytest.mark.parametrize("case", ["many32", "blob0", "blob1", "blob2", "blob3", "blob4", "blob5"])
def test_smooth_union_culling_changes_no_bit(case, monkeypatch):
    """CULL_LSE (a smooth-union child whose logsumexp term is exactly +0.0f for the whole wave is
    skipped) is an exact optimisation: the same scene compiled with RM_CULL_LSE=0 and with it gives identical values,
    point gradients, parameter gradients and frames -- through the interpreter, and (many32) through the specialised
    kernels, whose tape lives in LDS columns."""
    from ray_marching_amd import _abi, ops
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
    monkeypatch.setattr(ops, "bwd_hard_capacity", 0)      # bitwise parameter gradients: no atomically ordered deferred-ray list
    monkeypatch.setenv("RM_CULL_LSE_MIN", "2")
    monkeypatch.setenv("RM_SPECIALIZE", "off")
    gen = torch.Generator().manual_seed(4242 + sum(map(ord, case)))
    if case == "many32":
        make, spread = (lambda: make_many_primitive_scene(32)), 3.0
    else:
        spec, spread = H.random_blob_spec(gen, 8, 12)      # small enough for the interpreter's backward (accumulators in LDS)
        spec = O.map_spec(spec, lambda x: x.clone().float())
        make = lambda: H.spec_to_module(spec)
    pts = torch.cat([(torch.rand(64, 1, 3, generator=gen) * 2 - 1) * 2.5 * spread,
                     (torch.rand(64, 1, 3, generator=gen) * 2 - 1) * 0.5 * spread])
    pts = (pts + 0.02 * spread * torch.randn(128, 64, 3, generator=gen)).reshape(-1, 3).to(DEV)      # coherent waves
    wts = torch.randn(pts.shape[0], 1, generator=gen).to(DEV)
    same = lambda x, y: torch.equal(torch.nan_to_num(x, nan=1234.5), torch.nan_to_num(y, nan=1234.5)) \
        and torch.equal(x.isnan(), y.isnan())
    res = {}
    for path in (("off", "jit") if case == "many32" else ("off",)):      # jit: hipcc builds the culled program's library (~15 s)
        from ray_marching_amd import specialize
        if path == "jit" and not os.path.exists(specialize._hipcc()):
            continue
        monkeypatch.setenv("RM_SPECIALIZE", path)
        specialize._loaded.clear()
        for cull in ("0", "1"):
            monkeypatch.setenv("RM_CULL_LSE", cull)
            monkeypatch.setenv("RM_CULL_UNION_TABLE", cull)      # and the whole-union test from the children's own bounds
            module = make().to(DEV)
            cs = compiled_for(module)
            n_sites = int((cs.program.reshape(-1, 4)[:, 0] == _abi.OP_CULL_LSE).sum())
            assert (n_sites > 0) == (cull == "1")
            if path == "jit" and cull == "0":
                continue          # the unculled program through the specialised kernels is what every other test runs
            assert cs.specialised == (path == "jit")
            p = pts.clone().requires_grad_(True)
            d = module(p)
            (d * wts).sum().backward()
            gw = [None if x.grad is None else x.grad.clone() for x in module.parameters()]
            loop = H.make_loop(module, 40, 72)
            q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -1.5 * spread]], device=DEV)
            with torch.no_grad():
                frames = [loop(q, t, m, 1, 64) for m in (0, 4, 2, 5)]
            for prm in module.parameters():
                prm.grad = None
            loop(q, t, 0, 1, 24).pow(2).mean().backward()
            res[(path, cull)] = dict(d=d.detach(), gp=p.grad, frames=frames, gw=gw,
                                     gf=[None if x.grad is None else x.grad.clone() for x in module.parameters()])
    ref = res[("off", "0")]
    for key, got in res.items():
        if key == ("off", "0"):
            continue
        assert same(ref["d"], got["d"]) and same(ref["gp"], got["gp"]), key
        for x, y in zip(ref["frames"], got["frames"]):
            assert same(x, y), key
        if key[0] == "off":           # same kernels, same block size class: parameter gradients to summation order
            for name in ("gw", "gf"):
                for x, y in zip(ref[name], got[name]):
                    assert (x is None) == (y is None)
                    if x is not None:
                        scale = max(float(torch.nan_to_num(x).abs().max()), 1e-30)
                        assert torch.equal(x.isnan(), y.isnan())
                        assert float((torch.nan_to_num(x) - torch.nan_to_num(y)).abs().max()) <= 1e-4 * scale + 1e-9, (key, name)
    specialize._loaded.clear()


def test_config4_full_size_properties(monkeypatch):
    """BASELINE configs[3] at its real size -- closed make_test_scene, 512x512, 64 steps, Lambertian MSE (the shape the
    bench's fwd_bwd leg times; /root/reference/README.md:22-23, rendering/ray_marching.py:78-84) -- through the properties
    that do not need the oracle at 262 144 rays: the deferred-ray list (19.6 k rays here) against the in-place walk, a
    list far too small, HIP-graph replay against the eager step, and the LDS interpreter against the specialised
    kernels; all to summation order.  A 64x64 crop of the same frame is checked against the oracle's autograd."""
    from ray_marching_amd import ops, specialize
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.graphs import capture_step
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene
    h = w = 512
    steps = 64
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV)
    t = torch.tensor([[0.0, 0.0, -1.0]], device=DEV)
    target = torch.rand(1, h, w, 1, generator=torch.Generator().manual_seed(5)).to(DEV)

    def grads(capacity=None, path="auto", graph=False):
        monkeypatch.setenv("RM_SPECIALIZE", path)
        specialize._loaded.clear()
        ops.bwd_hard_capacity = capacity
        try:
            scene = make_closed_test_scene()
            loop = H.make_loop(scene, h, w)
            assert compiled_for(scene).specialised == (path == "auto")
            params = list(scene.parameters())

            def step():
                (loop(q, t, 0, 1, steps)[..., :1] - target).pow(2).mean().backward()

            if graph:
                g, _, _ = capture_step(step, params, warmup=2)
                for p in params:
                    p.grad.zero_()
                g.replay()
                torch.cuda.synchronize()
            else:
                ops.bwd_tile_cost_sink = torch.zeros(int(ops._lib.rm_wave_tiles(1, h, w, 2)), dtype=torch.int32, device=DEV)
                step()
                deferred = int(ops.bwd_last_work[32].item())
                grads.deferred = deferred
            return [p.grad.detach().clone() for p in params]
        finally:
            ops.bwd_hard_capacity = None
            ops.bwd_tile_cost_sink = None

    ref = grads(capacity=0)                         # every ray walked in place by its own wave
    assert grads.deferred == 0
    variants = {"default list": grads(), }
    n_def = grads.deferred
    assert 5000 < n_def < 60000, n_def              # ~19.6 k rays (7.5 %) are deferred at this shape
    variants["list of 1000 (overflowing)"] = grads(capacity=1000)
    variants["graph replay"] = grads(graph=True)
    variants["interpreter"] = grads(path="off")
    scale = max(float(g.abs().max()) for g in ref)
    for name, got in variants.items():
        for a, b in zip(ref, got):
            assert torch.isfinite(b).all(), name
            assert float((a - b).abs().max()) <= 2e-5 * max(scale, 1e-6) + 2e-6 * float(a.abs().max()), name
    print(f"config 4 at 512x512x64: {n_def} rays deferred; largest gradient component {scale:.3g}; 4 variants agree")
    specialize._loaded.clear()


def test_training_step_helper_matches_the_eager_loop():
    """RenderLoop.training_step: forward -> loss -> backward -> SGD captured into one HIP graph on first use.  Over five
    iterations with a pose that changes every iteration, each replayed step equals the eager step taken from the same
    parameters (loss and updated parameters to summation order; whole trajectories are not compared: one silhouette pixel
    flipping on a 1e-8 parameter difference moves this 64x64 MSE by 1e-4), and torch raises no stream-mismatch warning."""
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene
    h = w = 64
    target = torch.rand(1, h, w, 1, generator=torch.Generator().manual_seed(9)).to(DEV)
    loss_fn = lambda image: (image[..., :1] - target).pow(2).mean()
    poses = [(torch.nn.functional.normalize(torch.tensor([[1.0, 0.01 * i, -0.02 * i, 0.0]]), dim=-1).to(DEV),
              torch.tensor([[0.02 * i, 0.0, -1.0 - 0.05 * i]], device=DEV)) for i in range(5)]
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=".*AccumulateGrad node's stream does not match.*")
        scene = make_closed_test_scene()
        loop = H.make_loop(scene, h, w)
        opt = torch.optim.SGD(scene.parameters(), lr=1e-2)
        step = loop.training_step(loss_fn, mode=0, marching_steps=32, optimizer=opt)
        twin = make_closed_test_scene()
        twin_loop = H.make_loop(twin, h, w)
        twin_opt = torch.optim.SGD(twin.parameters(), lr=1e-2)
        step(*poses[0])                                  # first call: two warm-up iterations, the capture, one replay
        start = [p.detach().clone() for p in make_closed_test_scene().to(DEV).parameters()]
        assert any(float((a - b.detach()).abs().max()) > 0 for a, b in zip(start, scene.parameters())), "the optimiser is part of the graph"
        for q, t in poses:
            with torch.no_grad():
                for a, b in zip(twin.parameters(), scene.parameters()):
                    a.copy_(b)
            got_loss = float(step(q, t))
            twin_opt.zero_grad(set_to_none=True)
            want = loss_fn(twin_loop(q, t, 0, 1, 32))
            want.backward()
            twin_opt.step()
            assert abs(got_loss - float(want.detach())) <= 1e-6 * max(1.0, abs(float(want.detach()))), (got_loss, float(want.detach()))
            for a, b in zip(scene.parameters(), twin.parameters()):
                assert float((a - b).detach().abs().max()) <= 1e-6 * max(1.0, float(b.detach().abs().max()))


@pytest.mark.parametrize("mode", [1, 2, 5])
def test_shade_norm_backward_kernel_matches_the_tensor_formulas(mode):
    """rm_shade_norm_backward (two launches) against ops.minmax_normalisation_vjp / laplacian_normalisation_vjp (the
    same VJP written as ~15 tensor operations in the order autograd walks it): finite values to summation order,
    infinities and NaNs at exactly the same pixels -- on a frame with several pixels at the minimum and at the
    maximum, and on one whose upstream gradient is zero at the extremal pixels (0 * inf)."""
    from ray_marching_amd import _abi, ops
    gen = torch.Generator().manual_seed(31 + mode)
    n = 37 * 53
    for variant in range(3):
        raw = torch.randn(n, generator=gen) * (2.0 if mode == 5 else 0.5)
        if mode != 5:
            raw = raw - 1.0
        idx = torch.randperm(n, generator=gen)
        raw[idx[:3]] = raw.max() if mode != 5 else raw.abs().max() * (1.0 if variant else -1.0)   # several pixels at the extremum
        raw[idx[3:5]] = raw.min()
        g = torch.randn(n, 3, generator=gen)
        if variant == 2:
            g[idx[:5]] = 0.0
        raw3 = raw[:, None].expand(n, 3).contiguous().to(DEV)
        g = g.to(DEV)
        lo, hi = (raw.min(), raw.max()) if mode != 5 else (raw.min(), raw.abs().max())
        lohi = torch.stack([lo, hi]).to(DEV)
        want = ops.laplacian_normalisation_vjp(g, raw3[:, 0], lohi[1]) if mode == 5 \
            else ops.minmax_normalisation_vjp(g, raw3[:, 0], lohi[0], lohi[1])
        out = torch.empty_like(raw3)
        part = torch.empty(_abi.NORM_BWD_BLOCKS * 4, device=DEV)
        _abi.check(_abi.lib.rm_shade_norm_backward(_abi.ptr(raw3), _abi.ptr(g), _abi.ptr(lohi), mode, _abi.ptr(out),
                                                   _abi.ptr(part), n, _abi.current_stream(torch.device(DEV))), "rm_shade_norm_backward")
        got = out[:, 0]
        assert torch.equal(out[:, 1:], torch.zeros_like(out[:, 1:]))
        assert torch.equal(got.isnan(), want.isnan()) and torch.equal(got.isinf(), want.isinf()), (mode, variant)
        fin = torch.isfinite(want)
        assert torch.equal(torch.sign(got[~fin & ~want.isnan()]), torch.sign(want[~fin & ~want.isnan()]))
        scale = float(want[fin].abs().max()) if fin.any() else 1.0
        assert float((got[fin] - want[fin]).abs().max()) <= 2e-5 * scale + 1e-7, (mode, variant)
        assert int((~fin).sum()) > 0 or mode == 5          # the distance / proximity VJPs always carry non-finite entries


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_display_frame_is_the_three_pass_display_contract(dtype):
    """RenderLoop.display_frame: F.pad(images.mean(0).float(), [0,1], 1.0) (main.py:78-84; Window.draw's contiguous
    [H,W,4] fp32 input, torchwindow/window.py:146-174) written by the frame kernel itself -- bit-identical with the
    three tensor passes for all eight shader modes, fp32 and .half() modules, both frame kernels, and one launch."""
    import torch.nn.functional as F
    from torch.profiler import ProfilerActivity, profile
    h, w, steps = 90, 160, 32
    cmap = torch.from_numpy(H.gold("cmap.npz")["cyclic_cmap"])
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV).to(dtype)
    t = torch.tensor([[0.0, 0.0, 1.0]], device=DEV).to(dtype)
    for regen in (False, True):
        loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=regen)
        loop.shader.cyclic_cmap = cmap.to(DEV)
        loop = loop.to(dtype)                 # like main.py's .to(device, dtype): the colormap buffer is cast too
        for mode in range(8):
            with torch.no_grad():
                want = F.pad(loop(q, t, mode, 2, steps).mean(dim=0).float(), pad=[0, 1], value=1.0)
                got = loop.display_frame(q, t, mode, 2, steps)
            assert got.shape == (h, w, 4) and got.dtype == torch.float32 and got.is_contiguous()
            assert torch.equal(torch.nan_to_num(got, nan=-7.0), torch.nan_to_num(want, nan=-7.0)), (regen, mode)
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=False).to(dtype)
    with torch.no_grad():
        for _ in range(3):
            loop.display_frame(q, t, 4, 1, steps)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            loop.display_frame(q, t, 4, 1, steps)
            torch.cuda.synchronize()
    kernels = _device_kernels(prof)
    if kernels:
        others = [k for k in kernels if "k_render_fwd" not in k and "k_minmax_init" not in k and "Memset" not in k]
        assert not others, others
    # float64 colormap with an fp32 module (the reference's data file as loaded): the float64 product, then .float()
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w)
    loop.shader.cyclic_cmap = cmap.to(DEV)
    with torch.no_grad():
        for mode in (6, 7):
            img = loop(q.float(), t.float(), mode, 3, steps)
            assert img.dtype == torch.float64
            want = F.pad(img.mean(dim=0).float(), pad=[0, 1], value=1.0)
            assert torch.equal(torch.nan_to_num(loop.display_frame(q.float(), t.float(), mode, 3, steps), nan=-7.0), torch.nan_to_num(want, nan=-7.0))
    # the same tensor replayed from a HIP graph, pose changing between replays
    shot = loop.capture(4, 1, steps, display=True)
    for z in (1.0, -2.0):
        tz = torch.tensor([[0.0, 0.0, z]], device=DEV)
        with torch.no_grad():
            want = loop.display_frame(q.float(), tz, 4, 1, steps)
        assert torch.equal(shot(q.float(), tz), want)
    with pytest.raises(ValueError):
        H.make_loop(H.spec_to_module(O.scene_test2()), 16, 16, n=2).display_frame(torch.tensor([[1.0, 0, 0, 0]] * 2, device=DEV),
                                                                                  torch.zeros(2, 3, device=DEV))


@pytest.mark.parametrize("n_prims", [48, 100, 300])
def test_backward_of_wide_scenes(n_prims, monkeypatch):
    """Scenes far larger than the reference's own (scene_registry.py: <= 8 primitives; config 5: 32) train too: the
    generic backward keeps ONE accumulator row per wave in LDS (LdsStore::acc_row), so a smooth union of 300 affine-placed
    primitives (3080 parameters) gets its gradients through the same kernels.  Values bit-identical with the oracle in
    `restated` mode; dL/dp and dL/dtheta against the oracle's autograd, through `module(points)` and through a
    Lambertian frame with MSE loss (`k_render_bwd` + the deferred-ray kernels).  Round 2 / early round 3 refused the
    backward from 48 primitives on (accumulator columns per thread: "scene needs 171116 B of LDS")."""
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
    monkeypatch.setenv("RM_SPECIALIZE", "off")       # (the specialised library of such a scene also uses this backward)
    gen = torch.Generator().manual_seed(7 + n_prims)
    n_pts = 2048 if n_prims <= 100 else 512
    pts = (torch.rand(n_pts, 3, generator=gen) * 2 - 1) * 4.0
    w = torch.randn(n_pts, 1, generator=gen)
    mod = make_many_primitive_scene(n_prims).to(DEV)
    ref = O.map_spec(O.scene_many(n_prims), lambda x: x.clone().requires_grad_(True))
    named = O.spec_parameters(ref)
    assert [k for k, _ in named] == [k for k, _ in mod.named_parameters()]
    p = pts.to(DEV).requires_grad_(True)
    d = mod(p)
    with O.math_mode("restated"), torch.no_grad():
        assert torch.equal(d.detach().cpu(), O.sdf_eval(ref, pts))
    pc = pts.clone().requires_grad_(True)
    (d * w.to(DEV)).sum().backward()
    (O.sdf_eval(ref, pc) * w).sum().backward()
    assert float((p.grad.cpu() - pc.grad).abs().max()) <= 2e-5
    mine = torch.cat([x.grad.flatten().cpu() for x in mod.parameters()])
    theirs = torch.cat([x.grad.flatten() for _, x in named])
    assert mine.numel() == theirs.numel() >= 10 * n_prims
    assert float((mine - theirs).abs().max()) <= 1e-4 * float(theirs.abs().max()), "module(points) parameter gradients"
    if n_prims > 100:
        return                                       # (the oracle's frame autograd at 300 primitives takes minutes)
    for x in mod.parameters():
        x.grad = None
    for _, x in named:
        x.grad = None
    px = 3.45e-6
    loop = RenderLoop(mod, num_cameras=1, px_width=32, px_height=24, focal_length=px * 24, sensor_width=px * 32,
                      sensor_height=px * 24, normals_eps=H.EPS).to(DEV)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.0, 0.0, -4.0]])
    target = torch.rand(1, 24, 32, 1, generator=gen)
    img = loop(q.to(DEV), t.to(DEV), 0, 1, 32)
    (img[..., :1] - target.to(DEV)).pow(2).mean().backward()
    cam = O.camera_buffers(1, 32, 24, px * 24, px * 32, px * 24)
    want = O.render(ref, cam, q, t, 0, 1, 32, H.EPS)
    (want[..., :1] - target).pow(2).mean().backward()
    assert float((img.detach().cpu() - want.detach()).abs().max()) <= 1e-5
    mine = torch.cat([x.grad.flatten().cpu() for x in mod.parameters()])
    theirs = torch.cat([x.grad.flatten() for _, x in named])
    assert float((mine - theirs).abs().max()) <= 1e-4 * float(theirs.abs().max()), "frame parameter gradients"


def test_specialised_backward_with_row_accumulators(monkeypatch):
    """RM_STATIC_BACKWARD_ACC (opt-in): the per-scene library of a scene with more gradient accumulators than fit in
    registers (> 96) is built WITH its backward kernels, which then keep one accumulator row per wave in LDS
    (StaticCfg::kRowAcc).  Same gradients as the interpreter's backward, to summation order, through `module(points)`
    and through a frame; hipcc builds the library here (~1 min for 12 primitives)."""
    from ray_marching_amd import specialize
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
    if not os.path.exists(specialize._hipcc()):
        pytest.skip("no hipcc on this box")
    gen = torch.Generator().manual_seed(99)
    pts = ((torch.rand(4096, 3, generator=gen) * 2 - 1) * 4.0).to(DEV)
    w = torch.randn(4096, 1, generator=gen).to(DEV)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -4.0]], device=DEV)
    target = torch.rand(1, 48, 64, 1, generator=gen).to(DEV)
    got = {}
    for path in ("off", "jit"):
        monkeypatch.setenv("RM_SPECIALIZE", path)
        monkeypatch.setenv("RM_STATIC_BACKWARD_ACC", "400")
        specialize._loaded.clear()
        mod = make_many_primitive_scene(12).to(DEV)
        cs = compiled_for(mod)
        assert cs.n_params + cs.n_grad_derived > specialize.MAX_STATIC_BACKWARD_ACC and specialize.static_backward(cs)
        assert cs.specialised == (path == "jit")
        from ray_marching_amd import _abi
        assert (cs.lib(True) is not _abi.lib) == (path == "jit")        # the backward comes from the per-scene library
        p = pts.clone().requires_grad_(True)
        d = mod(p)
        (d * w).sum().backward()
        g1 = torch.cat([x.grad.flatten() for x in mod.parameters()])
        for x in mod.parameters():
            x.grad = None
        loop = H.make_loop(mod, 48, 64)
        img = loop(q, t, 0, 1, 64)
        (img[..., :1] - target).pow(2).mean().backward()
        g2 = torch.cat([x.grad.flatten() for x in mod.parameters()])
        got[path] = (d.detach(), p.grad, g1, img.detach(), g2)
    a, b = got["off"], got["jit"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
    assert float((a[1] - b[1]).abs().max()) <= 1e-6
    for i in (2, 4):
        assert float((a[i] - b[i]).abs().max()) <= 2e-5 * float(a[i].abs().max()), i


@pytest.mark.parametrize("mode", [0, 4, 6])
def test_two_camera_batch_trains(mode, monkeypatch):
    """num_cameras = 2 (control.py:201; main.py averages the batch for display) with everything requiring grad: scene
    parameters, both orientations and both translations -- `k_render_bwd`, the deferred-ray kernels and `k_camera_bwd`
    over a camera batch, against CPU autograd on the oracle.  Both deferred-ray settings (the list and the in-place walk)."""
    from ray_marching_amd import ops
    h, w, steps = 40, 48, 48
    spec = O.map_spec(O.scene_test1_closed(), lambda x: x.clone().requires_grad_(True))
    bufs = O.camera_buffers(2, w, h, H.PX * h, H.PX * w, H.PX * h)
    q0 = torch.nn.functional.normalize(torch.tensor([[0.98, 0.05, -0.12, 0.03], [0.95, -0.2, 0.1, 0.0]]), dim=-1)
    t0 = torch.tensor([[0.15, -0.1, -1.2], [-0.3, 0.2, -0.9]])
    gen = torch.Generator().manual_seed(21 + mode)
    wimg = torch.rand(2, h, w, 3, generator=gen)
    cmap = torch.from_numpy(H.gold("cmap.npz")["cyclic_cmap"]).float()
    qc, tc = q0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
    (O.render(spec, bufs, qc, tc, mode, 2, steps, H.EPS, cmap=cmap) * wimg).mean().backward()
    named = O.spec_parameters(spec)
    for cap in (None, 0):
        if cap is not None:
            monkeypatch.setattr(ops, "bwd_hard_capacity", cap)
        module = H.spec_to_module(O.scene_test1_closed())
        loop = H.make_loop(module, h, w, n=2)
        loop.shader.cyclic_cmap = cmap.to(DEV)
        qg, tg = q0.to(DEV).requires_grad_(True), t0.to(DEV).requires_grad_(True)
        img = loop(qg, tg, mode, 2, steps)
        assert img.shape == (2, h, w, 3)
        (img * wimg.to(DEV)).mean().backward()
        for name, got, want in (("orientation", qg.grad, qc.grad), ("translation", tg.grad, tc.grad)):
            assert got.shape == want.shape
            scale = max(1e-3, want.abs().max().item())
            err = (got.cpu() - want).abs().max().item()
            assert err <= 1e-4 * max(1.0, scale) and err <= 2e-3 * scale, (cap, name, err, scale)
        for (pname, want), (_, got) in zip(named, module.named_parameters()):
            assert (got.grad.cpu() - want.grad).abs().max().item() <= 1e-4, (cap, pname)


def test_backward_kernels_take_the_forward_scene_block(monkeypatch):
    """RmScene.block / block_out (ABI v12): block 0 of the recording forward leaves its finished scene block (parameters +
    derived constants), and k_render_bwd / k_bwd_hard_n / k_bwd_hard_b read it instead of gathering the parameters and
    deriving the constants again in every block (9-20 us per block for closed scene 1: the step 0.372 -> 0.340 ms at
    512^2).  Same gradients bit for bit as with the prologue run in every kernel; and the backward differentiates the
    frame that WAS rendered: parameters overwritten between forward and backward do not leak into it."""
    from ray_marching_amd import ops
    monkeypatch.setattr(ops, "bwd_hard_capacity", 0)          # in-place walk: bitwise reproducible parameter gradients
    h, w, steps = 96, 128, 64
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -1.0]], device=DEV)
    target = torch.rand(1, h, w, 1, generator=torch.Generator().manual_seed(5)).to(DEV)
    grads = {}
    for use in (False, True, "overwritten"):
        monkeypatch.setattr(ops, "use_forward_block", bool(use))
        module = H.spec_to_module(O.scene_test1_closed())
        loop = H.make_loop(module, h, w)
        loss = (loop(q, t, 0, 1, steps)[..., :1] - target).pow(2).mean()
        if use == "overwritten":
            saved = [p.detach().clone() for p in module.parameters()]
            with torch.no_grad():
                for p in module.parameters():
                    p.data.mul_(1.5)                           # (.data: no version bump, autograd cannot notice)
        loss.backward()
        grads[use] = torch.cat([p.grad.flatten() for p in module.parameters()])
    assert torch.equal(grads[False], grads[True])
    assert torch.equal(grads["overwritten"], grads[True])
    # ... and the pools' second kernel renders the same frame from the first one's block
    with torch.no_grad():
        frames = {}
        for use in (False, True):
            monkeypatch.setattr(ops, "use_forward_block", use)
            loop = H.make_loop(H.spec_to_module(O.scene_test2()), 90, 160, regen=True)
            frames[use] = [loop(q, torch.tensor([[0.0, 0.0, 1.0]], device=DEV), m, 1, 32) for m in (4, 0, 1)]
        for a, b in zip(frames[False], frames[True]):
            assert torch.equal(a, b)


def test_scene_cache_is_checked_against_the_live_parameters(monkeypatch):
    """RmScene.block_cache (ABI v13): inference launches reuse the derived constants of the previous launch on their
    stream ONLY when the parameters they gather are bit for bit the ones the cache belongs to.  Frames with the cache and
    without are identical; an in-place edit, a `.data` write, an optimiser-style update and a value that moves a cull
    bound are all seen by the very next frame (both kernel paths, both frame kernels); NaN parameters never match a
    stale cache by accident; two streams keep separate caches."""
    from ray_marching_amd import ops
    from ray_marching_amd.compiler import compiled_for
    h, w, steps = 72, 96, 48
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -3.0]], device=DEV)

    def edits(module):
        prm = dict(module.named_parameters())
        names = list(prm)
        yield "unchanged", lambda: None
        yield "in-place", lambda: prm[names[0]].data.mul_(1.25)
        yield ".data assignment", lambda: setattr(prm[names[1]], "data", prm[names[1]].data + 0.125)
        yield "unchanged again", lambda: None
        yield "every parameter", lambda: [p.data.add_(0.01 * torch.randn_like(p)) for p in prm.values()]
        yield "back to a value seen before (the cache holds the latest only)", lambda: prm[names[0]].data.div_(1.25)

    for path in ("off", "auto"):
        monkeypatch.setenv("RM_SPECIALIZE", path)
        for regen in (False, True):
            torch.manual_seed(3)
            cached_mod, plain_mod = H.spec_to_module(O.scene_test2()), H.spec_to_module(O.scene_test2())
            cached, plain = H.make_loop(cached_mod, h, w, regen=regen), H.make_loop(plain_mod, h, w, regen=regen)
            for (what, edit_a), (_, edit_b) in zip(edits(cached_mod), edits(plain_mod)):
                torch.manual_seed(11); edit_a()
                torch.manual_seed(11); edit_b()
                with torch.no_grad():
                    monkeypatch.setattr(ops, "use_scene_cache", True)
                    a = [cached(q, t, m, 1, steps) for m in (4, 0)] + [cached(q, t, 4, 1, steps)]       # second frame of a value: a hit
                    monkeypatch.setattr(ops, "use_scene_cache", False)
                    b = [plain(q, t, m, 1, steps) for m in (4, 0)] + [plain(q, t, 4, 1, steps)]
                for x, y in zip(a, b):
                    assert torch.equal(torch.nan_to_num(x, nan=-7.0), torch.nan_to_num(y, nan=-7.0)), (path, regen, what)
            monkeypatch.setattr(ops, "use_scene_cache", True)
            assert compiled_for(cached_mod).__dict__.get("_block_caches"), "the cache was never created"
            assert not compiled_for(plain_mod).__dict__.get("_block_caches")
    # a NaN parameter: frames equal the uncached ones (NaN != NaN never lets a stale cache through: the compare is on bits,
    # and the bits changed)
    monkeypatch.setenv("RM_SPECIALIZE", "off")
    mod_a, mod_b = H.spec_to_module(O.scene_test2()), H.spec_to_module(O.scene_test2())
    la, lb = H.make_loop(mod_a, h, w), H.make_loop(mod_b, h, w)
    with torch.no_grad():
        la(q, t, 4, 1, steps)
        for m in (mod_a, mod_b):
            next(iter(m.parameters())).data.fill_(float("nan"))
        monkeypatch.setattr(ops, "use_scene_cache", True)
        xa = la(q, t, 4, 1, steps)
        monkeypatch.setattr(ops, "use_scene_cache", False)
        xb = lb(q, t, 4, 1, steps)
    assert torch.equal(xa.isnan(), xb.isnan()) and torch.equal(torch.nan_to_num(xa, nan=-7.0), torch.nan_to_num(xb, nan=-7.0))
    # a second stream gets its own cache
    monkeypatch.setattr(ops, "use_scene_cache", True)
    mod = H.spec_to_module(O.scene_test2())
    loop = H.make_loop(mod, h, w)
    side = torch.cuda.Stream()
    with torch.no_grad():
        x0 = loop(q, t, 4, 1, steps)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            x1 = loop(q, t, 4, 1, steps)
        torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(x0, x1) and len(compiled_for(mod).__dict__["_block_caches"]) == 2
