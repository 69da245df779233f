"""HIP kernels (through the C ABI) vs the committed reference outputs and vs the oracle.

Tolerance (north_star): forward pixels within 1e-5 fp32.  The kernels follow the reference's ATen op order
with FMA contraction off; pow and atan2 restate the Sleef functions ATen calls (tests/test_math_sweep.py), so
everything without a smooth union is asserted BIT-EXACT, all eight shader modes included.  torch.logsumexp
goes through MKL VML exp/log, whose bits depend on the host CPU (profiles/host_math_probe.py): there the
device computes the correctly rounded value, and the asserts are "within 1 ulp per node" / 1e-5 per pixel.
"""
import os

import numpy as np
import pytest
import torch

from oracle import sdf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL = 1e-5


@pytest.fixture(params=["generic", "static"])
def kernel_path(request, monkeypatch):
    """Run a test through the LDS interpreter ('generic') and through the per-scene
    compile-time specialised kernels ('static', prebuilt by __graft_entry__.build())."""
    from ray_marching_amd import specialize
    monkeypatch.setenv("RM_SPECIALIZE", "off" if request.param == "generic" else "auto")
    specialize._loaded.clear()
    if request.param == "static" and not getattr(kernel_path, "_built", False):
        specialize.prebuild_default_scenes()      # no-op when __graft_entry__.build() already made them
        kernel_path._built = True
    yield request.param
    specialize._loaded.clear()


def _check_path(module, kernel_path):
    from ray_marching_amd.compiler import compiled_for
    cs = compiled_for(module)
    assert cs.specialised == (kernel_path == "static"), \
        "specialised library missing: run __graft_entry__.build()" if kernel_path == "static" else "generic expected"


def test_extension_loaded():
    from ray_marching_amd import _abi
    assert _abi.lib.rm_abi_version() == _abi.ABI_VERSION


NODES = ["sphere", "box", "plane", "line", "disk", "torus", "affine", "rounding", "onion", "union",
         "smooth_union", "scene1", "scene2", "scene1_closed", "scene_many8"]


@pytest.mark.parametrize("name", NODES)
def test_nodes_vs_golden(name):
    g = H.gold("f1_nodes.npz")
    pts = torch.from_numpy(g["points"]).to(DEV)
    module = H.spec_to_module(H.node_specs()[name]).to(DEV)
    with torch.no_grad():
        got = module(pts)
    assert got.shape == (pts.shape[0], 1)
    mx, frac = H.report(name, got, g[name])
    ulps = H.ulp_distance(got, g[name])
    print(f"{name}: max|err|={mx:.3g}, {int((ulps > 0).sum())} of {ulps.numel()} values differ, worst {int(ulps.max())} ulp")
    if name in ("smooth_union", "scene1", "scene1_closed", "scene_many8"):
        # logsumexp: MKL's exp/log (fixture host) vs the correctly rounded ones, <= 1 ulp each, 1.5 % / 0.01 % of inputs
        assert int(ulps.max()) <= 1 and float((ulps > 0).double().mean()) <= 0.01, (name, int(ulps.max()))
    else:
        assert mx == 0.0, f"{name} expected bit-exact, got {mx}"


def test_leading_shapes_and_empty():
    module = H.spec_to_module(O.scene_test2()).to(DEV)
    x = torch.randn(2, 5, 7, 3, device=DEV)
    with torch.no_grad():
        d = module(x)
        assert d.shape == (2, 5, 7, 1)
        assert torch.equal(d.reshape(-1, 1), module(x.reshape(-1, 3)))
        assert module(torch.empty(0, 3, device=DEV)).shape == (0, 1)
    with pytest.raises(ValueError):
        module(torch.zeros(4, 2, device=DEV))
    with pytest.raises(RuntimeError):
        module(torch.zeros(4, 3))  # CPU tensor: no fallback


def test_camera_vs_golden():
    from ray_marching_amd.rendering.ray_marching import PinholeCamera
    g = H.gold("f2_camera.npz")
    h, w = (int(x) for x in g["hw"])
    cam = PinholeCamera(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    assert np.array_equal(cam.ray_positions.numpy(), g["ray_positions"])
    assert np.array_equal(cam.ray_directions.numpy(), g["ray_directions"])
    cam = cam.to(DEV)
    for i in (0, 1):
        pos, frames, pos2, dirs = cam(torch.from_numpy(g[f"q{i}"]).to(DEV), torch.from_numpy(g[f"t{i}"]).to(DEV))
        assert pos2 is pos
        for name, got in (("pos", pos), ("dirs", dirs), ("frames", frames)):
            mx, _ = H.report(name, got, g[f"{name}{i}"])
            assert mx == 0.0, (name, i, mx)
    cam2 = PinholeCamera(2, w, h, H.PX * h, H.PX * w, H.PX * h).to(DEV)
    pos, frames, _, dirs = cam2(torch.from_numpy(g["q_n2"]).to(DEV), torch.from_numpy(g["t_n2"]).to(DEV))
    for name, got in (("pos", pos), ("dirs", dirs), ("frames", frames)):
        assert H.report(name, got, g[f"{name}_n2"])[0] == 0.0


def test_config1_sphere_distance_shader(kernel_path):
    """BASELINE config 1: SDFSphere(0.5), 256x256, 32 steps, depth (distance) shader."""
    g = H.gold("f3_sphere.npz")
    h, w = (int(x) for x in g["hw"])
    loop = H.make_loop(H.spec_to_module(O.scene_sphere(0.5)), h, w)
    _check_path(loop.scene, kernel_path)
    q, t = torch.from_numpy(g["q"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    with torch.no_grad():
        img = loop(q, t, 1, 1, int(g["steps"]))
    assert img.shape == (1, h, w, 3)
    s = int(g["stride"])
    mx, frac = H.report("image", img[:, ::s, ::s, :1], g["image_sub"])
    print(f"config1 distance image: max|err|={mx:.3g}")
    assert mx <= TOL
    assert abs(img[..., 0].double().mean().item() - float(g["image_mean"])) < 1e-6
    assert torch.equal(img[..., 0], img[..., 1]) and torch.equal(img[..., 0], img[..., 2])


FRAMES = ["f4_scene2_64_s32_in.npz", "f4_scene2_64_s128_out.npz", "f4_scene2_90x160_s128_tilt.npz",
          "f4_scene1c_64_s64.npz"]


@pytest.mark.parametrize("name", FRAMES)
@pytest.mark.parametrize("early", [True, False])
def test_frames_vs_golden(name, early, kernel_path):
    g = H.gold(name)
    h, w = (int(x) for x in g["hw"])
    spec = O.scene_test1_closed() if "scene1c" in name else O.scene_test2()
    loop = H.make_loop(H.spec_to_module(spec), h, w, early_out=early)
    _check_path(loop.scene, kernel_path)
    loop.shader.cyclic_cmap = torch.from_numpy(H.gold("cmap.npz")["cyclic_cmap"]).to(DEV)
    q, t = torch.from_numpy(g["q"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    degree = int(g["degree"]) if "degree" in g.files else 1
    steps = int(g["steps"])
    exact_scene = "scene1c" not in name
    for key in g.files:
        if not key.startswith("mode"):
            continue
        m = int(key[4:])
        with torch.no_grad():
            img = loop(q, t, m, degree, steps)
        want = g[key]
        assert img.shape == (1, h, w, 3)
        got = img[..., : want.shape[-1]]
        mx, frac = H.report(f"{name} mode{m}", got, want)
        print(f"{name} mode {m} early={early}: max|err|={mx:.3g} frac>1e-5={frac:.3g}")
        assert mx <= TOL, (name, m, mx)
        if m in (6, 7):
            # fp32 brightness * float64 colormap row = float64 image (shader.py:104,118).  The colormap index
            # (Sleef atan2) is exact; the brightness .pow(1/2) (shader.py:116) is MKL vsSqrt in ATen, which is
            # not correctly rounded (0.6 % of inputs 1 ulp off, tests/golden/math_sweep.json) -- the kernel's
            # IEEE sqrt may therefore sit one fp32 ulp of the brightness away, nothing more
            assert img.dtype == torch.float64 and mx <= 1.2e-7, (name, m, mx)
        elif exact_scene:
            # the image itself: pow is Sleef's bits, the distance shaders' log lands on MKL's value for these pixels
            assert mx == 0.0, f"mode {m} on scene2 expected bit-exact, got {mx}"
    # intermediate tensors through the stand-alone modules (marcher / normals / scene call)
    with torch.no_grad():
        pos, frames, _, dirs = loop.camera(q, t)
        p = loop.marcher(pos, dirs, steps)
        n, lap = loop.normals(p)
        dist = loop.scene(p)
    tight = 0.0 if exact_scene else 1e-6   # smooth-union scene: exp/log are MKL's on the fixture host (<= 1 ulp away)
    errs = {"p": H.report("p", p, g["p"])[0], "n": H.report("n", n, g["n"])[0]}
    if "dist" in g.files:
        errs["dist"] = H.report("dist", dist, g["dist"])[0]
        errs["lap"] = H.report("lap", lap, g["lap"])[0]
    print(f"{name} intermediates: " + " ".join(f"{k}={v:.3g}" for k, v in errs.items()))
    assert errs["p"] <= tight and errs["n"] <= (0.0 if exact_scene else TOL)
    if "dist" in errs:
        # Laplacian = (centre - mean of 4 taps) * 6/eps^2 = 2400 x a difference of distances: 1 ulp of one tap is 1.4e-4
        assert errs["dist"] <= tight and errs["lap"] <= (0.0 if exact_scene else 5e-4)


@pytest.mark.parametrize("loss_name,mode", [("lambert_mse", 0), ("normal_sq", 4)])
def test_backward_vs_golden(loss_name, mode, kernel_path):
    """Config 4 shape: grads of every scene parameter through the fused frame, vs the
    reference's autograd (fp32 fixture; the fp64 fixture bounds the reference's own rounding)."""
    g = H.gold("f5_backward.npz")
    h, w = (int(x) for x in g["hw"])
    module = H.spec_to_module(O.scene_test1_closed())
    loop = H.make_loop(module, h, w)
    _check_path(module, kernel_path)
    q, t = torch.from_numpy(g["q"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    img = loop(q, t, mode, 1, int(g["steps"]))
    mx, _ = H.report("image", img[..., : (1 if mode == 0 else 3)], g[f"{loss_name}_image"])
    assert mx <= TOL
    if loss_name == "lambert_mse":
        loss = (img[..., :1] - torch.from_numpy(g["target"]).to(DEV)).pow(2).mean()
    else:
        loss = img.pow(2).mean()
    loss.backward()
    assert abs(loss.item() - float(g[f"{loss_name}_f32_loss"])) < 1e-6
    worst = 0.0
    for pname, prm in module.named_parameters():
        want32 = torch.from_numpy(g[f"{loss_name}_f32_grad:{pname}"])
        want64 = torch.from_numpy(g[f"{loss_name}_f64_grad:{pname}"])
        got = prm.grad.cpu()
        err32 = (got - want32).abs().max().item()
        err64 = (got.double() - want64).abs().max().item()
        ref_noise = (want32.double() - want64).abs().max().item()
        worst = max(worst, min(err32, err64))
        print(f"{loss_name} {pname}: |hip-ref32|={err32:.2e} |hip-ref64|={err64:.2e} |ref32-ref64|={ref_noise:.2e}")
        assert min(err32, err64) <= 1e-4, (pname, err32, err64)   # north_star: grads within 1e-4
    print(f"{loss_name}: worst grad error {worst:.3g}")


def test_tie_subgradients_vs_golden():
    g = H.gold("f6_ties.npz")
    cases = {
        "union_tie": ("union", {}, [O.scene_sphere(0.5), O.scene_sphere(0.5)]),
        "box_face": ("box", {"halfsides": O._t((0.5, 0.5, 0.5))}),
        "line_clamp": ("line", {"start": O._t((0.0, 0.0, 0.0)), "end": O._t((1.0, 0.0, 0.0)), "radius": O._t(0.1)}),
        "onion_zero": ("onion", {"radius": O._t(0.1)}, O.scene_sphere(1.0)),
        "smooth_tie": ("smooth_union", {"blend_k": O._t(22.0)}, [O.scene_sphere(0.5), O.scene_sphere(0.5)]),
        "disk_edge": ("disk", {"radius": O._t(0.8)}),
    }
    for name, spec in cases.items():
        module = H.spec_to_module(spec).to(DEV)
        p = torch.from_numpy(g[name + "_points"]).to(DEV).requires_grad_(True)
        d = module(p)
        d.sum().backward()
        assert H.report(name + " d", d, g[name + "_d"])[0] <= 1e-6
        assert H.report(name + " grad_p", p.grad, g[name + "_grad_p"])[0] <= 1e-6, name
        for pname, prm in module.named_parameters():
            assert H.report(f"{name} {pname}", prm.grad, g[f"{name}_grad:{pname}"])[0] <= 2e-6, (name, pname)


@pytest.mark.parametrize("name", ["scene1_closed", "scene2", "scene_many8", "disk", "rounding"])
def test_sdf_backward_vs_oracle_autograd(name):
    """grad w.r.t. query points and every parameter on seeded points, vs CPU autograd on the oracle."""
    spec = O.map_spec(H.node_specs()[name], lambda x: x.clone().requires_grad_(True))
    module = H.spec_to_module(spec).to(DEV)
    gen = torch.Generator().manual_seed(7)
    pts = (torch.rand(3000, 3, generator=gen) * 6 - 3)
    w = torch.randn(3000, 1, generator=gen)
    p_cpu = pts.clone().requires_grad_(True)
    (O.sdf_eval(spec, p_cpu) * w).sum().backward()
    p_gpu = pts.to(DEV).requires_grad_(True)
    (module(p_gpu) * w.to(DEV)).sum().backward()
    assert H.report("grad_points", p_gpu.grad, p_cpu.grad)[0] <= 1e-5
    for (pname, want), (_, got) in zip(O.spec_parameters(spec), module.named_parameters()):
        scale = max(1.0, want.grad.abs().max().item())
        err = (got.grad.cpu() - want.grad).abs().max().item()
        assert err <= 1e-4 * scale, (pname, err, scale)


def test_march_and_normals_backward_vs_oracle(kernel_path):
    spec = O.map_spec(O.scene_test1_closed(), lambda x: x.clone().requires_grad_(True))
    module = H.spec_to_module(spec).to(DEV)
    _check_path(module, kernel_path)
    from ray_marching_amd.rendering.ray_marching import SDFMarcher, SDFNormals
    h, w, steps = 24, 32, 40
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.1, 0.0, -1.0]])
    pos, _, dirs = O.camera_forward(*bufs, q, t)
    gen = torch.Generator().manual_seed(3)
    wn = torch.randn(1, h, w, 3, generator=gen); wl = torch.randn(1, h, w, 1, generator=gen) * 1e-3
    wp = torch.randn(1, h, w, 3, generator=gen)
    # oracle
    pos_c, dirs_c = pos.clone().requires_grad_(True), dirs.clone().requires_grad_(True)
    p = O.march(spec, pos_c, dirs_c, steps)
    n, lap = O.normals(spec, p, H.EPS)
    ((n * wn).sum() + (lap * wl).sum() + (p * wp).sum()).backward()
    # HIP
    pos_g, dirs_g = pos.to(DEV).requires_grad_(True), dirs.to(DEV).requires_grad_(True)
    pg = SDFMarcher(module)(pos_g, dirs_g, steps)
    ng, lg = SDFNormals(module, H.EPS).to(DEV)(pg)
    assert H.report("p", pg, p)[0] <= 2e-6
    ((ng * wn.to(DEV)).sum() + (lg * wl.to(DEV)).sum() + (pg * wp.to(DEV)).sum()).backward()
    for name, got, want in (("grad_pos", pos_g.grad, pos_c.grad), ("grad_dirs", dirs_g.grad, dirs_c.grad)):
        scale = max(1.0, want.abs().max().item())
        mx = H.report(name, got, want)[0]
        print(f"{name}: max|err|={mx:.3g} (scale {scale:.3g})")
        assert mx <= 1e-4 * scale, (name, mx)
    for (pname, want), (_, got) in zip(O.spec_parameters(spec), module.named_parameters()):
        scale = max(1.0, want.grad.abs().max().item())
        err = (got.grad.cpu() - want.grad).abs().max().item()
        assert err <= 1e-4 * scale, (pname, err, scale)


def test_open_scene_nan_pattern_matches():
    """rays that miss an open scene blow up; the inf/NaN pattern must match."""
    spec = O.scene_sphere(0.5)
    h = w = 48
    loop = H.make_loop(H.spec_to_module(spec), h, w)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.0, 0.0, -2.0]])
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    for steps in (32, 128):
        with torch.no_grad():
            want, aux = O.render(spec, bufs, q, t, 4, 1, steps, H.EPS, return_aux=True)
            got = loop(q.to(DEV), t.to(DEV), 4, 1, steps)
        mx, frac = H.report(f"open sphere S={steps}", got, want)   # asserts identical NaN positions
        assert mx <= TOL


def test_full_size_properties(kernel_path):
    """BASELINE config 2 size (1920x1080x128): properties that need no oracle run."""
    h, w, steps = 1080, 1920, 128
    loop_e = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, early_out=True)
    _check_path(loop_e.scene, kernel_path)
    loop_f = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, early_out=False)
    loop_t = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, early_out=True, tile8x8=True)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -3.0]], device=DEV)
    with torch.no_grad():
        for mode in (0, 4):
            a = loop_e(q, t, mode, 1, steps)
            b = loop_f(q, t, mode, 1, steps)
            c = loop_t(q, t, mode, 1, steps)
            assert torch.equal(a, b), "early-out changed pixels"
            assert torch.equal(a, c), "8x8 wave tiling changed pixels"
            assert torch.isfinite(a).all()                      # closed scene: every ray hits
            # row tiles reassemble to the full frame bit-for-bit (multi-GPU sharding property)
            tiles = [loop_e(q, t, mode, 1, steps, rows=(r, r + 135)) for r in range(0, h, 135)]
            assert torch.equal(torch.cat(tiles, dim=1), a)
        nrm = loop_e(q, t, 4, 1, steps)
        assert float(nrm.min()) >= 0.0 and float(nrm.max()) <= 1.0
        # unit normals: |n|^2 == 1 within rounding where no clamp was hit
        sq = nrm.pow(2).sum(-1)
        assert (sq - 1).abs().max().item() < 1e-5
        # the centre 64x64 crop equals an oracle render of the same pixels
        bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
        crop = tuple(b[:, 508:572, 928:992].contiguous() for b in bufs)
        want = O.render(O.scene_test2(), crop, q.cpu(), t.cpu(), 4, 1, steps, H.EPS)
        assert H.report("centre crop", nrm[:, 508:572, 928:992], want)[0] <= TOL


def test_static_and_generic_agree_bitwise(monkeypatch):
    """The specialised kernels run the same handlers as the interpreter: identical bits, on the
    32-primitive smooth-union scene of config 5 (forward) and on the closed scene-1 (gradients)."""
    from ray_marching_amd import specialize
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene, make_many_primitive_scene
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -4.5]], device=DEV)
    outs, grads = {}, {}
    for path in ("off", "auto"):
        monkeypatch.setenv("RM_SPECIALIZE", path)
        specialize._loaded.clear()
        loop = H.make_loop(make_many_primitive_scene(32), 96, 128)
        assert compiled_for(loop.scene).specialised == (path == "auto")
        with torch.no_grad():
            outs[path] = [loop(q, t, m, 2, 96) for m in (0, 4, 1)]
        scene = make_closed_test_scene()
        loop = H.make_loop(scene, 48, 48)
        loop(q, torch.tensor([[0.0, 0.0, -1.0]], device=DEV), 0, 1, 48).pow(2).mean().backward()
        grads[path] = [p.grad.clone() for p in scene.parameters()]
    specialize._loaded.clear()
    for a, b in zip(outs["off"], outs["auto"]):
        assert torch.equal(a, b)
    for a, b in zip(grads["off"], grads["auto"]):
        # per-block partial sums are grouped differently (LDS columns vs wave butterflies)
        assert (a - b).abs().max().item() <= 1e-6 * max(1.0, a.abs().max().item())


def test_config5_scene_vs_golden_and_host_spread():
    """32-primitive smooth-union scene (config 5 shape, small frame) against the reference's output from the
    build container (fixture f9).  Two legs:

    * `restated` -- the oracle with exp / log of oracle/rm_math_ref.c (two restatements by the same author, tied to
      torch through the exhaustive sweep's <= 1-ulp counts, tests/test_math_sweep.py): the device must agree bit for
      bit at every pixel.  That pins everything but MKL's ulps: 128 steps x 32 affine primitives, ATen's summation order
      inside logsumexp, the culling, the early-out.
    * the reference fixture itself (MKL VML exp / log, CPU-dispatched): every device value is within north_star's 1e-5
      of it, OR the pixel is one the reference ITSELF cannot resolve to 1e-5 -- its own float32 and float64 renders
      of the frame (both in the fixture) differ by more than 1e-5 there: a ray grazing a blend crease, where one ulp
      of exp() decides which object it ends on.  The pixels that may differ are thus named by the reference, not by
      a flat budget; their number is printed.  The same rule for p_final.
    Next to it, how far the SAME reference arithmetic lands from the fixture on this box's CPU (informational)."""
    from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
    g = H.gold("f9_many32_54x96_s128.npz")
    h, w = (int(x) for x in g["hw"]); steps = int(g["steps"])
    loop = H.make_loop(make_many_primitive_scene(32), h, w)
    q, t = torch.from_numpy(g["q"]), torch.from_numpy(g["t"])
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    spec = O.scene_many(32)
    tol = 1e-5

    def excused_or_close(got, want32, want64, what):
        """per element: |got - want32| <= tol, or the reference's own fp32-vs-fp64 spread at that PIXEL exceeds tol"""
        got = got.detach().cpu().double().numpy()
        err = np.abs(got - want32.astype(np.float64))
        spread = np.abs(want32.astype(np.float64) - want64).max(axis=-1, keepdims=True)     # per pixel
        ill = np.broadcast_to(spread > tol, err.shape)
        assert np.isfinite(got).all() and np.isfinite(want64).all(), what
        off = err > tol
        print(f"config5 {what}: max|err|={err.max():.3g}; {int(off.sum())} of {err.size} values beyond {tol:g}, "
              f"{int((off & ~ill).sum())} of them on pixels the reference resolves; reference fp32-vs-fp64 spread > {tol:g} "
              f"on {int((spread > tol).sum())} of {spread.size} pixels")
        assert not (off & ~ill).any(), (what, float(err[off & ~ill].max()))
        return int(off.sum())

    with torch.no_grad():
        pos, frames, _, dirs = loop.camera(q.to(DEV), t.to(DEV))
        excused_or_close(loop.marcher(pos, dirs, steps), g["p"], g["p_f64"], "p_final")
    for mode in (0, 4):
        with torch.no_grad():
            host = O.render(spec, bufs, q, t, mode, 1, steps, H.EPS)
            with O.math_mode("restated"):      # same op stream, exp/log of oracle/rm_math_ref.c: host independent
                exact = O.render(spec, bufs, q, t, mode, 1, steps, H.EPS)
            got = loop(q.to(DEV), t.to(DEV), mode, 1, steps)
        assert H.report(f"config5 mode {mode} restated", got, exact)[0] == 0.0
        want = g[f"mode{mode}"]
        ch = want.shape[-1]
        excused_or_close(got[..., :ch], want, g[f"mode{mode}_f64"], f"mode {mode}")
        hmx, hfrac = H.report(f"config5 mode {mode} host", host[..., :ch], want)
        print(f"config5 scene mode {mode}: this host's reference arithmetic vs fixture max|err|={hmx:.3g} frac>1e-5={hfrac:.3g}")


def test_two_camera_batch_vs_golden(kernel_path):
    """num_cameras = 2 (the reference renders [N,H,W,3] and main.py averages over N)."""
    g = H.gold("f8_scene2_two_cameras.npz")
    h, w = (int(x) for x in g["hw"])
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, n=2)
    _check_path(loop.scene, kernel_path)
    q, t = torch.from_numpy(g["q"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    for m in (0, 1, 4):
        with torch.no_grad():
            img = loop(q, t, m, 2, int(g["steps"]))
        assert img.shape == (2, h, w, 3)
        want = g[f"mode{m}"]
        mx, _ = H.report(f"two cameras mode {m}", img[..., : want.shape[-1]], want)
        assert mx == 0.0, (m, mx)
    with pytest.raises(ValueError):
        loop(q[:1], t[:1], 0, 1, 8)       # pose batch must match num_cameras


def test_fp16_io_config3_numerics():
    """Config 3 numerics: module.to(float16) = fp16 buffers/parameters/output, fp32 arithmetic inside
    the kernel.  The reference's fp16 path rounds every op to fp16 and itself deviates from its fp32
    path by up to 0.66 in p (fixture ref_spread_*), so parity is statistical: we must be within ~1e-2
    (p99) of the fp16 reference and CLOSER to the fp32 reference than the fp16 reference is."""
    g = H.gold("f7_scene2_fp16_90x160_s32.npz")
    h, w = (int(x) for x in g["hw"])
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w).to(torch.float16)
    q = torch.from_numpy(g["q"]).to(DEV).half(); t = torch.from_numpy(g["t"]).to(DEV).half()
    for m, key in ((0, "mode0"), (4, "mode4")):
        with torch.no_grad():
            img = loop(q, t, m, 1, int(g["steps"]))
        assert img.dtype == torch.float16 and img.shape == (1, h, w, 3)
        mine = img[..., : g[key + "_16"].shape[-1]].float().cpu().numpy()
        d16 = np.abs(mine - g[key + "_16"]); d32 = np.abs(mine - g[key + "_32"])
        ref = np.abs(g[key + "_16"] - g[key + "_32"])
        print(f"fp16 {key}: vs ref16 p99={np.quantile(d16, .99):.3g} frac>1e-2={np.mean(d16 > 1e-2):.3g}; "
              f"vs ref32 p99={np.quantile(d32, .99):.3g}; ref16 vs ref32 p99={np.quantile(ref, .99):.3g}")
        assert np.quantile(d16, 0.99) <= 2e-2 and np.mean(d16 > 1e-2) <= 0.025
        assert np.quantile(d32, 0.99) <= np.quantile(ref, 0.99)        # fp32 arithmetic: closer to the fp32 truth
        assert np.median(d16) <= 1e-3


def test_config4_pose_optimisation():
    """BASELINE config 4 (reduced to 128x128 so the CPU check stays in seconds): gradients of the
    MSE-to-target loss w.r.t. the 3 translations + 3 quaternions match CPU autograd on the oracle
    within 1e-4 (the contract), and a short Adam run on those gradients finds a lower loss (the exact
    gradient of the discrete render is badly conditioned -- see examples/optimize_scene.py -- so only
    "some descent" is asserted)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "optimize_scene.py")
    spec_ = importlib.util.spec_from_file_location("optimize_scene", path)
    ex = importlib.util.module_from_spec(spec_); spec_.loader.exec_module(ex)
    size, steps = 128, 64
    loop, target_loop = ex.make_problem(size, DEV)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -1.0]], device=DEV)
    with torch.no_grad():
        target = target_loop(q, t, 0, 1, steps)[..., :1]
    loss = (loop(q, t, 0, 1, steps)[..., :1] - target).pow(2).mean()
    loss.backward()
    # same problem on the oracle (CPU autograd)
    spec = O.map_spec(O.scene_test1_closed(), lambda x: x.clone().requires_grad_(True))
    bufs = O.camera_buffers(1, size, size, H.PX * size, H.PX * size, H.PX * size)
    img = O.render(spec, bufs, q.cpu(), t.cpu(), 0, 1, steps, H.EPS)[..., :1]
    loss_cpu = (img - target.cpu()).pow(2).mean()
    loss_cpu.backward()
    assert abs(loss.item() - loss_cpu.item()) <= 1e-6
    worst = 0.0
    for (name, want), (_, got) in zip(O.spec_parameters(spec), loop.scene.named_parameters()):
        err = (got.grad.cpu() - want.grad).abs().max().item()
        worst = max(worst, err)
        assert err <= 1e-4, (name, err)
    print(f"config 4 (128^2 x 64): loss {loss.item():.4e}, worst |grad error| vs CPU autograd {worst:.2e}")
    r = ex.run(size=128, march_steps=64, iters=150, lr=1e-3, device=DEV, log=lambda *_: None)
    print(f"config 4 optimisation: loss {r['loss_first']:.3e} -> best {min(r['losses']):.3e}")
    assert min(r["losses"]) < r["loss_first"]


@pytest.mark.parametrize("mode", [0, 4])
def test_camera_pose_gradients(mode, kernel_path):
    """d loss / d (orientations, translations) through the fused frame and through the
    stand-alone camera -> marcher -> normals chain, vs CPU autograd on the oracle."""
    h, w, steps = 40, 48, 48
    spec = O.scene_test1_closed()
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    q0 = torch.nn.functional.normalize(torch.tensor([[0.98, 0.05, -0.12, 0.03]]), dim=-1)
    t0 = torch.tensor([[0.15, -0.1, -1.2]])
    gen = torch.Generator().manual_seed(11)
    wimg = torch.rand(1, h, w, 3, generator=gen)
    qc, tc = q0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
    (O.render(spec, bufs, qc, tc, mode, 1, steps, H.EPS) * wimg).mean().backward()
    # fused frame
    module = H.spec_to_module(spec)
    loop = H.make_loop(module, h, w)
    _check_path(module, kernel_path)
    qg, tg = q0.to(DEV).requires_grad_(True), t0.to(DEV).requires_grad_(True)
    (loop(qg, tg, mode, 1, steps) * wimg.to(DEV)).mean().backward()
    for name, got, want in (("orientation", qg.grad, qc.grad), ("translation", tg.grad, tc.grad)):
        scale = max(1e-3, want.abs().max().item())
        err = (got.cpu() - want).abs().max().item()
        print(f"fused mode {mode} grad {name}: |err|={err:.2e} (scale {scale:.2e})")
        assert err <= 1e-4 * max(1.0, scale) and err <= 2e-3 * scale, (name, err, scale)
    # parameter grads are still right when the pose also requires grad
    spec_g = O.map_spec(spec, lambda x: x.clone().requires_grad_(True))
    (O.render(spec_g, bufs, q0, t0, mode, 1, steps, H.EPS) * wimg).mean().backward()
    for (pname, want), (_, got) in zip(O.spec_parameters(spec_g), module.named_parameters()):
        assert (got.grad.cpu() - want.grad).abs().max().item() <= 1e-4, pname
    # stand-alone chain: camera -> marcher -> normals -> (torch) lambert-like reduction
    q2, t2 = q0.to(DEV).requires_grad_(True), t0.to(DEV).requires_grad_(True)
    pos, frames, _, dirs = loop.camera(q2, t2)
    p = loop.marcher(pos, dirs, steps)
    n, lap = loop.normals(p)
    ((n * dirs).sum(-1, keepdim=True) * wimg.to(DEV)[..., :1]).mean().backward()
    q3, t3 = q0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
    pos_c, _, dirs_c = O.camera_forward(*bufs, q3, t3)
    p_c = O.march(spec, pos_c, dirs_c, steps)
    n_c, _ = O.normals(spec, p_c, H.EPS)
    ((n_c * dirs_c).sum(-1, keepdim=True) * wimg[..., :1]).mean().backward()
    for name, got, want in (("orientation", q2.grad, q3.grad), ("translation", t2.grad, t3.grad)):
        scale = max(1e-3, want.abs().max().item())
        err = (got.cpu() - want).abs().max().item()
        assert err <= 1e-4 * max(1.0, scale) and err <= 2e-3 * scale, ("chain", name, err, scale)


@pytest.mark.parametrize("shape", [(1, 1, 1), (1, 3, 5), (1, 9, 7), (3, 10, 13), (2, 8, 8), (1, 17, 130)])
@pytest.mark.parametrize("tile8", [True, False])
def test_ragged_frames_vs_oracle(shape, tile8, kernel_path):
    """Frame sizes that are not multiples of the 8x8 wave tile / 64-ray wave, several cameras, both
    wave->pixel mappings, whole frame and an odd row band; steps = 0 and steps = 37."""
    n, h, w = shape
    spec = O.scene_test2()
    loop = H.make_loop(H.spec_to_module(spec), h, w, n=n, tile8x8=tile8)
    _check_path(loop.scene, kernel_path)
    gen = torch.Generator().manual_seed(n * 1000 + h * 10 + w)
    q = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0, 0.0, 0.0]]) + 0.1 * torch.randn(n, 4, generator=gen), dim=-1)
    t = torch.tensor([[0.0, 0.0, -3.0]]) + 0.2 * torch.randn(n, 3, generator=gen)
    bufs = O.camera_buffers(n, w, h, H.PX * h, H.PX * w, H.PX * h)
    for steps in (0, 37):
        for mode in (4, 0):
            with torch.no_grad():
                want = O.render(spec, bufs, q, t, mode, 1, steps, H.EPS)
                got = loop(q.to(DEV), t.to(DEV), mode, 1, steps)
            assert got.shape == (n, h, w, 3)
            assert H.report(f"{shape} S={steps} mode {mode}", got, want)[0] == 0.0
            if h >= 3:
                band = loop(q.to(DEV), t.to(DEV), mode, 1, steps, rows=(1, h - 1))
                assert torch.equal(band, got[:, 1:h - 1])
    # one globally normalised mode: min/max over ALL cameras (shader.py:35-36)
    if n * h * w > 1:
        with torch.no_grad(), O.math_mode("restated"):
            # the oracle's log from oracle/rm_math_ref.c (correctly rounded): this box's MKL log sits 1 ulp away
            # on ~1e-5 of inputs, which x^(1/2.33) would amplify next to the global minimum.  The fixtures from the
            # build container (test_frames_vs_golden) pin the same shader to the reference's own bits.
            want = O.render(spec, bufs, q, t, 1, 1, 24, H.EPS)
            got = loop(q.to(DEV), t.to(DEV), 1, 1, 24)
        assert H.report(f"{shape} distance", got, want)[0] == 0.0


def test_bad_calls_raise():
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), 8, 8)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.zeros(1, 3, device=DEV)
    from ray_marching_amd._abi import RmError
    with pytest.raises(RmError):
        loop(q, t, 0, 1, 8, rows=(4, 4))          # empty band
    with pytest.raises(RmError):
        loop(q, t, 0, 1, 8, rows=(0, 9))          # band outside the frame
    with pytest.raises(RmError):
        loop(q, t, 0, 1, -1)                      # negative step count
    with pytest.raises(NotImplementedError):
        loop.shader(None, q, None, None, None, None, None, None, mode="lambertian", degree=1)


def test_captured_graph_replay_matches_eager():
    """RenderLoop.capture(): HIP-graph replay of a frame equals the eager call, follows pose changes and
    in-place scene-parameter edits without re-capture."""
    h, w, steps = 90, 160, 48
    module = H.spec_to_module(O.scene_test2())
    loop = H.make_loop(module, h, w)
    q = torch.nn.functional.normalize(torch.tensor([[0.95, 0.05, 0.25, -0.1]]), dim=-1).to(DEV)
    for mode in (4, 1):
        frame = loop.capture(mode, 1, steps)
        for z in (-3.0, -2.0):
            t = torch.tensor([[0.2, 0.0, z]], device=DEV)
            with torch.no_grad():
                want = loop(q, t, mode, 1, steps)
            assert torch.equal(frame(q, t), want)
        with torch.no_grad():
            module.sdfs[1].sdfs[0].radius.add_(0.2)          # edit the sphere radius in place
            want = loop(q, t, mode, 1, steps)
        assert torch.equal(frame(q, t), want)
        with torch.no_grad():
            module.sdfs[1].sdfs[0].radius.sub_(0.2)


def test_standalone_chain_with_shader_modules_is_differentiable():
    """camera -> marcher -> normals -> Shader(mode) composed by hand like RenderLoop.forward
    (control.py:239-257), back-propagated to scene parameters and pose, vs CPU autograd."""
    h, w, steps = 24, 32, 40
    spec = O.map_spec(O.scene_test1_closed(), lambda x: x.clone().requires_grad_(True))
    module = H.spec_to_module(spec)
    loop = H.make_loop(module, h, w)
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    q0 = torch.nn.functional.normalize(torch.tensor([[0.98, 0.05, -0.12, 0.03]]), dim=-1)
    t0 = torch.tensor([[0.15, -0.1, -1.2]])
    gen = torch.Generator().manual_seed(5)
    for mode in (0, 3, 4):
        ch = 3 if mode == 4 else 1
        wimg = torch.rand(1, h, w, ch, generator=gen)
        for _, p in O.spec_parameters(spec):
            p.grad = None
        for p in module.parameters():
            p.grad = None
        qc, tc = q0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
        (O.render(spec, bufs, qc, tc, mode, 1, steps, H.EPS)[..., :ch] * wimg).mean().backward()
        qg, tg = q0.to(DEV).requires_grad_(True), t0.to(DEV).requires_grad_(True)
        pos, frames, _, dirs = loop.camera(qg, tg)
        p = loop.marcher(pos, dirs, steps)
        dist = loop.scene(p)
        n, lap = loop.normals(p)
        img = loop.shader(pos, qg, frames, dirs, p, n, lap, dist, mode=mode, degree=1)
        assert img.shape == (1, h, w, ch)
        (img * wimg.to(DEV)).mean().backward()
        for name, got, want in (("orientation", qg.grad, qc.grad), ("translation", tg.grad, tc.grad)):
            if mode == 3 and name == "orientation":
                continue      # frames (QuaternionToSO3) is not differentiated here; dirs carries the pose
            if want is None:  # vignette does not depend on the camera translation at all
                assert got is None or float(got.abs().max()) == 0.0
                continue
            scale = max(1e-3, want.abs().max().item())
            err = (got.cpu() - want).abs().max().item()
            assert err <= 1e-4 * max(1.0, scale) and err <= 2e-3 * scale, (mode, name, err, scale)
        if mode != 3:
            for (pname, want), (_, got) in zip(O.spec_parameters(spec), module.named_parameters()):
                assert (got.grad.cpu() - want.grad).abs().max().item() <= 1e-4, (mode, pname)


@pytest.mark.parametrize("cam_z", [-3.0, 1.0])
def test_config2_whole_frame_sample_vs_oracle(cam_z):
    """BASELINE config 2 at full size (1920x1080, 128 steps, scene2): every 8th pixel row and column
    of the complete frame -- walls with their sliding rays, silhouettes, the torus interior for the
    reference's default camera (0,0,1) -- must equal the oracle bit for bit (normal + Lambertian)."""
    h, w, steps, stride = 1080, 1920, 128, 8
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.0, 0.0, cam_z]])
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    sub = tuple(b[:, ::stride, ::stride].contiguous() for b in bufs)
    for mode in (4, 0):
        with torch.no_grad():
            want = O.render(O.scene_test2(), sub, q, t, mode, 1, steps, H.EPS)
            got = loop(q.to(DEV), t.to(DEV), mode, 1, steps)[:, ::stride, ::stride]
        mx, _ = H.report(f"config 2 z={cam_z} mode {mode}", got, want)
        assert mx == 0.0, (cam_z, mode, mx)


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_trees_vs_oracle(seed):
    """Random compositions of all 11 node types (nested affine / smooth-union / union / rounding /
    onion, un-normalised quaternions, 1-child unions ...) through the interpreter: values at 1500
    points and the gradients w.r.t. points and every parameter, vs the oracle."""
    gen = torch.Generator().manual_seed(1000 + seed)
    spec = O.map_spec(H.random_spec(gen), lambda x: x.clone().float().requires_grad_(True))
    module = H.spec_to_module(spec).to(DEV)
    pts = torch.rand(1500, 3, generator=gen) * 5 - 2.5
    wts = torch.randn(1500, 1, generator=gen)
    p_cpu = pts.clone().requires_grad_(True)
    d_cpu = O.sdf_eval(spec, p_cpu)
    (d_cpu * wts).sum().backward()
    p_gpu = pts.to(DEV).requires_grad_(True)
    d_gpu = module(p_gpu)
    (d_gpu * wts.to(DEV)).sum().backward()
    exact = not H.spec_has(spec, "smooth_union")
    mx, _ = H.report(f"tree {seed}", d_gpu, d_cpu)
    assert mx <= (0.0 if exact else 2e-6), (seed, mx)       # aten mode: this host's MKL exp/log in the smooth unions
    spec_ng = O.map_spec(spec, lambda x: x.detach())
    with torch.no_grad(), O.math_mode("restated"):            # host-independent exp/log: bit-exact for every tree
        assert H.report(f"tree {seed} restated", d_gpu, O.sdf_eval(spec_ng, pts))[0] == 0.0
    gscale = max(1.0, p_cpu.grad.abs().max().item())
    assert H.report("grad points", p_gpu.grad, p_cpu.grad)[0] <= 2e-5 * gscale
    for (pname, want), (_, got) in zip(O.spec_parameters(spec), module.named_parameters()):
        if want.grad is None:
            assert got.grad is None or float(got.grad.abs().max()) == 0.0
            continue
        scale = max(1.0, want.grad.abs().max().item())
        err = (got.grad.cpu() - want.grad).abs().max().item()
        assert err <= 2e-4 * scale, (seed, pname, err, scale)
    # and a short march + normals through the same tree (kinks, NaN-free or NaN-identical)
    from ray_marching_amd.rendering.ray_marching import SDFMarcher, SDFNormals
    o = torch.tensor([[0.0, 0.0, -4.0]]).expand(256, 3).contiguous()
    v = torch.nn.functional.normalize(torch.rand(256, 3, generator=gen) - torch.tensor([0.5, 0.5, -0.5]), dim=-1)
    with torch.no_grad(), O.math_mode("restated"):
        p_ref = O.march(spec_ng, o, v, 12)
        n_ref, lap_ref = O.normals(spec_ng, p_ref, H.EPS)
        p_got = SDFMarcher(module)(o.to(DEV), v.to(DEV), 12)
        n_got, lap_got = SDFNormals(module, H.EPS).to(DEV)(p_got)
    assert H.report("march", p_got, p_ref)[0] == 0.0
    assert H.report("normals", n_got, n_ref)[0] == 0.0
    assert H.report("laplacian", lap_got, lap_ref)[0] == 0.0


@pytest.mark.parametrize("case", ["scene2", "scene1c", "many32"] + [f"tree{i}" for i in (4, 11, 13, 14, 17, 18, 21)])
def test_union_culling_changes_no_bit(case, monkeypatch):
    """CULL_MIN (bounding-sphere skip of min-union children) is an exact optimisation:
    the same scene compiled with RM_CULL=0 and with culling gives identical values, identical
    point gradients and identical parameter gradients, and identical frames."""
    from ray_marching_amd import _abi, ops
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.scene.scene_registry import (make_closed_test_scene, make_many_primitive_scene,
                                                       make_test_scene2)
    monkeypatch.setenv("RM_SPECIALIZE", "off")          # same (interpreter) kernels on both sides
    # bitwise comparison of parameter gradients: keep every ray in its own wave (the deferred-ray list is filled
    # through an atomic counter, so its order -- and with it the grouping of partial sums -- varies run to run)
    monkeypatch.setattr(ops, "bwd_hard_capacity", 0)
    gen = torch.Generator().manual_seed(77)
    pts = torch.cat([torch.rand(4096, 3, generator=gen) * 8 - 4,          # far from the objects: culls fire
                     torch.rand(4096, 3, generator=gen) * 2 - 1]).to(DEV)   # among them: they do not
    wts = torch.randn(8192, 1, generator=gen).to(DEV)

    def make():
        if case == "scene2":
            return make_test_scene2()
        if case == "many32":
            return make_many_primitive_scene(32)
        if case == "scene1c":                   # evaluation order differs from child order; |q|^2 = 1.00002
            return make_closed_test_scene()
        g = torch.Generator().manual_seed(1000 + int(case[4:]))
        return H.spec_to_module(O.map_spec(H.random_spec(g), lambda x: x.clone().float()))

    res, n_cull = {}, {}
    monkeypatch.setenv("RM_CULL_MIN_COST", "40" if not case.startswith("tree") else "0")   # trees: cull everything boundable
    for cull in ("0", "1"):
        monkeypatch.setenv("RM_CULL", cull)
        module = make().to(DEV)
        cs = compiled_for(module)
        n_cull[cull] = int((cs.program.reshape(-1, 4)[:, 0] == _abi.OP_CULL_MIN).sum())
        p = pts.clone().requires_grad_(True)
        d = module(p)
        (d * wts).sum().backward()
        loop = H.make_loop(module, 40, 72)
        q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -3.0]], device=DEV)
        with torch.no_grad():
            frames = [loop(q, t, m, 1, 48) for m in (0, 4, 5)]
        for prm in module.parameters():
            prm.grad = None
        loop(q, t, 0, 1, 24).pow(2).mean().backward()
        res[cull] = dict(d=d.detach(), gp=p.grad, frames=frames,
                         gw=[None if x.grad is None else x.grad.clone() for x in module.parameters()])
    assert n_cull["0"] == 0
    assert n_cull["1"] > 0, "every case here is expected to carry CULL_MIN instructions"
    a, b = res["0"], res["1"]
    same = lambda x, y: torch.equal(torch.nan_to_num(x, nan=1234.5), torch.nan_to_num(y, nan=1234.5)) \
        and torch.equal(x.isnan(), y.isnan())
    assert same(a["d"], b["d"]) and same(a["gp"], b["gp"])
    for x, y in zip(a["frames"], b["frames"]):
        assert same(x, y)
    for x, y in zip(a["gw"], b["gw"]):
        assert (x is None) == (y is None)
        if x is not None:
            assert same(x, y)


@pytest.mark.parametrize("case", ["scene2", "scene1c", "many32"])
def test_tracked_culls_in_specialised_kernels_change_no_bit(case, monkeypatch):
    """The specialised march loop carries cull decisions from step to step (Scene::eval_near: bounds on
    the test value moved by |f||v|, refreshed every 16 steps).  Frames from those kernels equal the frames
    of the interpreter running the same scene compiled WITHOUT any CULL_MIN, bit for bit, from cameras
    inside the object group, far outside it, and looking along a wall."""
    from ray_marching_amd import specialize
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.scene.scene_registry import (make_closed_test_scene, make_many_primitive_scene,
                                                       make_test_scene2)
    make = {"scene2": make_test_scene2, "scene1c": make_closed_test_scene,
            "many32": lambda: make_many_primitive_scene(32)}[case]
    poses = [([1.0, 0.0, 0.0, 0.0], [0.0, 0.0, -3.0]), ([1.0, 0.0, 0.0, 0.0], [0.0, 0.0, 1.0]),
             ([0.9239, 0.0, 0.3827, 0.0], [4.5, -4.0, -4.5]), ([0.7071, 0.7071, 0.0, 0.0], [0.3, 4.7, 0.2])]
    frames = {}
    for variant in ("interp_nocull", "static_tracked"):
        monkeypatch.setenv("RM_CULL", "0" if variant == "interp_nocull" else "1")
        monkeypatch.setenv("RM_SPECIALIZE", "off" if variant == "interp_nocull" else "auto")
        specialize._loaded.clear()
        loop = H.make_loop(make(), 72, 128)
        assert compiled_for(loop.scene).specialised == (variant == "static_tracked")
        out = []
        with torch.no_grad():
            for q, t in poses:
                qq = torch.tensor([q], device=DEV); tt = torch.tensor([t], device=DEV)
                out += [loop(qq, tt, m, 1, 160) for m in (4, 0, 5)]
        frames[variant] = out
    specialize._loaded.clear()
    for a, b in zip(frames["interp_nocull"], frames["static_tracked"]):
        assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))
        assert torch.equal(a.isnan(), b.isnan())


@pytest.mark.parametrize("case,seed", [("scene1c", 0), ("scene1c", 1), ("many32", 2), ("scene2", 3)])
def test_cull_bounds_follow_the_live_parameters(case, seed, monkeypatch):
    """The cull bounds are derived on the device from the parameters of THIS launch.  Scenes whose objects
    were moved against the walls, resized and given far-from-unit quaternions (the state a scene is in
    half-way through an optimisation) still render bit-identically with the specialised, culling,
    decision-carrying kernels and with the interpreter compiled without CULL_MIN."""
    from ray_marching_amd import specialize
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.scene.scene_registry import (make_closed_test_scene, make_many_primitive_scene,
                                                       make_test_scene2)
    make = {"scene2": make_test_scene2, "scene1c": make_closed_test_scene,
            "many32": lambda: make_many_primitive_scene(32)}[case]
    gen = torch.Generator().manual_seed(4242 + seed)
    state = {}
    for name, p in make().named_parameters():
        x = p.detach().clone()
        if name.endswith("translation") or name.endswith("start") or name.endswith("end"):
            x = x + (torch.rand(x.shape, generator=gen) * 6 - 3)
        elif name.endswith("orientation"):
            x = (x + torch.randn(x.shape, generator=gen) * 0.3) * (0.8 + 0.5 * torch.rand(1, generator=gen))
        elif name.endswith("blend_k"):
            x = (x * (0.3 + 1.5 * torch.rand(1, generator=gen))).reshape(x.shape)
        elif "sdfs.0." in name and case != "scene1c" or name == "sdfs.1.sdf.halfsides" or name == "sdfs.1.radius":
            pass                                         # keep the room where it is
        else:
            x = x * (0.4 + 1.6 * torch.rand(x.shape, generator=gen))
        state[name] = x
    poses = [([1.0, 0.0, 0.0, 0.0], [0.0, 0.0, -3.0]), ([0.9239, 0.0, 0.3827, 0.0], [4.0, -3.5, -4.0])]
    frames = {}
    for variant in ("interp_nocull", "static_tracked"):
        monkeypatch.setenv("RM_CULL", "0" if variant == "interp_nocull" else "1")
        monkeypatch.setenv("RM_SPECIALIZE", "off" if variant == "interp_nocull" else "auto")
        specialize._loaded.clear()
        scene = make()
        with torch.no_grad():
            for name, p in scene.named_parameters():
                p.copy_(state[name])
        loop = H.make_loop(scene, 64, 96)
        assert compiled_for(loop.scene).specialised == (variant == "static_tracked")
        out = []
        with torch.no_grad():
            for q, t in poses:
                qq = torch.tensor([q], device=DEV); tt = torch.tensor([t], device=DEV)
                out += [loop(qq, tt, m, 1, 96) for m in (4, 0)]
        frames[variant] = out
    specialize._loaded.clear()
    for a, b in zip(frames["interp_nocull"], frames["static_tracked"]):
        assert torch.equal(a.isnan(), b.isnan())
        assert torch.equal(torch.nan_to_num(a, nan=7.0), torch.nan_to_num(b, nan=7.0))


def test_training_step_captured_in_a_graph_matches_eager():
    """forward + backward of the fused frame (config 4 shape, small) recorded with torch.cuda.graph:
    the replayed gradients equal the eager ones (up to the grouping of per-block partial sums, which
    follows the dynamic tile schedule) and track parameter updates made between replays."""
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene
    scene = make_closed_test_scene()
    loop = H.make_loop(scene, 64, 64)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -1.0]], device=DEV)
    target = torch.rand(1, 64, 64, 1, device=DEV)
    params = list(scene.parameters())

    def step():
        loss = (loop(q, t, 0, 1, 32)[..., :1] - target).pow(2).mean()
        loss.backward()
        return loss

    def eager():
        for p in params:
            p.grad = None
        loss = step()
        return loss.item(), [p.grad.clone() for p in params]

    import warnings
    from ray_marching_amd.graphs import capture_step
    with warnings.catch_warnings():
        # For the WHOLE body, the eager steps after the capture included: warm-up and capture run on one stream
        # (graphs.py), and the captured result comes back detached, so no AccumulateGrad node of the capture's side
        # stream survives to meet a later eager backward -- torch must not warn anywhere here.
        warnings.filterwarnings("error", message=".*AccumulateGrad node's stream does not match.*")
        graph, loss, captured = capture_step(step, params, warmup=2)
        assert loss.grad_fn is None
        for trial in range(2):
            graph.replay()
            torch.cuda.synchronize()
            got_loss, got = loss.item(), [g.clone() for g in captured]
            want_loss, want = eager()
            assert abs(got_loss - want_loss) <= 1e-6
            for a, b in zip(got, want):
                assert (a - b).abs().max().item() <= 1e-6 * max(1.0, b.abs().max().item())
            with torch.no_grad():                       # move the scene; the graph reads the live parameters
                for p in params:
                    p.add_(0.01 * torch.randn_like(p))


def test_fast_precision_is_within_tolerance_of_the_reference():
    """Opt-in precision="fast" (1-ulp v_sqrt_f32, reciprocal normalise, FMA contraction; still fp32).
    Not bit-exact by construction and NOT the product default: >= 99.8 % of pixel values stay within 1e-5
    (a 1-ulp change can move a silhouette ray to another object), but parameter gradients of the
    ill-conditioned config-4 loss move by ~3e-4 -- outside north_star's 1e-4, inside the reference's own
    fp32-vs-fp64 spread (~1e-3) -- which is why the exact build is the default and the headline."""
    g = H.gold("f4_scene2_90x160_s128_tilt.npz")
    h, w = (int(x) for x in g["hw"])
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, precision="fast")
    from ray_marching_amd.compiler import compiled_for
    assert compiled_for(loop.scene).lib(False, "fast") is not compiled_for(loop.scene).lib(False, "exact")
    q, t = torch.from_numpy(g["q"]).to(DEV), torch.from_numpy(g["t"]).to(DEV)
    for m in (0, 4):
        with torch.no_grad():
            img = loop(q, t, m, 1, int(g["steps"]))
        want = g[f"mode{m}"]
        mx, frac = H.report(f"fast mode {m}", img[..., : want.shape[-1]], want)
        print(f"fast arithmetic, mode {m}: max|err|={mx:.3g}, fraction of values beyond 1e-5: {frac:.3g}")
        assert frac <= 2e-3, (m, mx, frac)
    # whole 1080p frame sample vs the oracle
    hh, ww, steps, stride = 1080, 1920, 128, 8
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), hh, ww, precision="fast")
    qq = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); tt = torch.tensor([[0.0, 0.0, -3.0]])
    bufs = O.camera_buffers(1, ww, hh, H.PX * hh, H.PX * ww, H.PX * hh)
    sub = tuple(b[:, ::stride, ::stride].contiguous() for b in bufs)
    with torch.no_grad():
        want = O.render(O.scene_test2(), sub, qq, tt, 4, 1, steps, H.EPS)
        got = loop(qq.to(DEV), tt.to(DEV), 4, 1, steps)[:, ::stride, ::stride]
    mx, frac = H.report("fast 1080p", got, want)
    print(f"fast arithmetic, 1080p sample: max|err|={mx:.3g}, fraction beyond 1e-5: {frac:.3g}")
    assert frac <= 2e-3
    # gradients (config-4 shape fixture)
    gb = H.gold("f5_backward.npz")
    h, w = (int(x) for x in gb["hw"])
    module = H.spec_to_module(O.scene_test1_closed())
    loop = H.make_loop(module, h, w, precision="fast")
    q, t = torch.from_numpy(gb["q"]).to(DEV), torch.from_numpy(gb["t"]).to(DEV)
    img = loop(q, t, 0, 1, int(gb["steps"]))
    (img[..., :1] - torch.from_numpy(gb["target"]).to(DEV)).pow(2).mean().backward()
    worst = max((p.grad.cpu() - torch.from_numpy(gb[f"lambert_mse_f32_grad:{n}"])).abs().max().item()
                for n, p in module.named_parameters())
    ref_spread = max(float(np.abs(gb[f"lambert_mse_f32_grad:{n}"] - gb[f"lambert_mse_f64_grad:{n}"]).max())
                     for n, _ in module.named_parameters())
    print(f"fast arithmetic: worst parameter-gradient error {worst:.3g} (reference fp32-vs-fp64 spread {ref_spread:.3g})")
    assert worst <= ref_spread and worst <= 1e-3


@pytest.mark.parametrize("seed", [4, 17, 21])
def test_jit_specialised_random_trees_match_interpreter(seed, monkeypatch, tmp_path):
    """The compile-time specialisation of an arbitrary topology (nested unions / smooth unions / affine
    chains; built here with hipcc, ~10 s each) must reproduce the interpreter bit for bit: values, a
    short frame, and (to summation order) gradients."""
    from ray_marching_amd import _abi, specialize
    from ray_marching_amd.compiler import compile_scene, compiled_for
    if specialize._hipcc() is None or not os.path.exists(specialize._hipcc()):
        pytest.skip("hipcc not available on this box")
    monkeypatch.setattr(specialize, "SPEC_DIR", str(tmp_path))
    gen = torch.Generator().manual_seed(1000 + seed)
    spec = O.map_spec(H.random_spec(gen), lambda x: x.clone().float())
    results = {}
    for policy in ("off", "jit"):
        monkeypatch.setenv("RM_SPECIALIZE", policy)
        specialize._loaded.clear()
        module = H.spec_to_module(spec).to(DEV)
        cs = compiled_for(module)
        assert cs.specialised == (policy == "jit")
        pts = (torch.rand(4096, 3, generator=torch.Generator().manual_seed(seed)) * 5 - 2.5).to(DEV).requires_grad_(True)
        d = module(pts)
        d.sum().backward()
        loop = H.make_loop(module, 40, 56)
        q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -4.0]], device=DEV)
        with torch.no_grad():
            img = loop(q, t, 4, 1, 24)
        results[policy] = (d.detach(), pts.grad.clone(), img, [p.grad.clone() for p in module.parameters()])
    specialize._loaded.clear()
    a, b = results["off"], results["jit"]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(torch.nan_to_num(a[2], nan=-7.0), torch.nan_to_num(b[2], nan=-7.0))
    for ga, gb in zip(a[3], b[3]):
        assert (ga - gb).abs().max().item() <= 1e-5 * max(1.0, ga.abs().max().item())


def test_module_surface_edge_cases():
    """Broadcast shapes, half-precision stand-alone modules, a parameter-free scene and shared module
    instances (one nn.Parameter used by two nodes), all vs the oracle."""
    from ray_marching_amd.rendering.ray_marching import SDFMarcher
    from ray_marching_amd.scene.primitives import SDFPlane, SDFSphere
    from ray_marching_amd.scene.transformations import SDFAffineTransformation, SDFUnion
    gen = torch.Generator().manual_seed(9)
    # 1. marcher broadcasting: one origin [1,1,3] against directions [H,W,3]; gradient sums back to [1,1,3]
    spec = O.map_spec(O.scene_test2(), lambda x: x.clone())
    module = H.spec_to_module(spec).to(DEV)
    dirs = torch.nn.functional.normalize(torch.randn(6, 9, 3, generator=gen) + torch.tensor([0.0, 0.0, 3.0]), dim=-1)
    o_c = torch.tensor([[[0.1, -0.2, -3.0]]], requires_grad=True)
    p_c = O.march(spec, o_c, dirs, 20)
    p_c.sum().backward()
    o_g = o_c.detach().to(DEV).requires_grad_(True)
    p_g = SDFMarcher(module)(o_g, dirs.to(DEV), 20)
    assert p_g.shape == (6, 9, 3) and H.report("broadcast march", p_g, p_c)[0] == 0.0
    p_g.sum().backward()
    assert o_g.grad.shape == (1, 1, 3)
    assert (o_g.grad.cpu() - o_c.grad).abs().max().item() <= 1e-4 * max(1.0, o_c.grad.abs().max().item())
    # 2. half-precision stand-alone call: fp16 in, fp16 out, fp32 arithmetic inside
    x16 = (torch.rand(257, 3, generator=gen) * 4 - 2).half().to(DEV)
    with torch.no_grad():
        d16 = module.half()(x16)
        want = O.sdf_eval(O.map_spec(spec, lambda t: t.half().float()), x16.float().cpu())
    assert d16.dtype == torch.float16 and torch.equal(d16.cpu(), want.half())   # fp16 storage, fp32 arithmetic, one rounding
    # 3. a scene without parameters
    plane = SDFPlane().to(DEV)
    x = torch.randn(100, 3, generator=gen).to(DEV).requires_grad_(True)
    d = plane(x)
    d.sum().backward()
    assert torch.equal(d.detach()[:, 0], x.detach()[:, 0]) and torch.equal(x.grad, torch.tensor([1.0, 0.0, 0.0], device=DEV).expand(100, 3))
    # 4. the same SDFSphere instance under two parents: its radius receives both contributions
    ball = SDFSphere(0.4)
    shared = SDFUnion([ball, SDFAffineTransformation(ball, orientation=[1.0, 0.0, 0.0, 0.0], translation=[1.5, 0.0, 0.0])]).to(DEV)
    r = torch.tensor(0.4, requires_grad=True); tr = torch.tensor([1.5, 0.0, 0.0], requires_grad=True)
    qq = torch.tensor([1.0, 0.0, 0.0, 0.0], requires_grad=True)
    sspec = ("union", {}, [("sphere", {"radius": r}), ("affine", {"translation": tr, "orientation": qq}, ("sphere", {"radius": r}))])
    pts = torch.rand(500, 3, generator=gen) * 4 - 1
    wts = torch.randn(500, 1, generator=gen)
    (O.sdf_eval(sspec, pts) * wts).sum().backward()
    (shared(pts.to(DEV)) * wts.to(DEV)).sum().backward()
    assert len(list(shared.parameters())) == 3          # radius, translation, orientation (radius only once)
    assert abs(ball.radius.grad.item() - r.grad.item()) <= 1e-4 * max(1.0, abs(r.grad.item()))
    assert (shared.sdfs[1].translation.grad.cpu() - tr.grad).abs().max().item() <= 1e-4


@pytest.mark.parametrize("capacity", [None, 37])
def test_deferred_rays_give_the_same_gradients(capacity, kernel_path):
    """rm_render_backward hands rays whose march has not settled to the (ray, step)-parallel kernels
    k_bwd_hard_n/_a/_b.  Scene-parameter and pose gradients equal the ones of the in-place walk (list switched
    off) up to summation order -- with the default list, and with a list far too small (37 rays: most are
    walked in place, the rest deferred; both paths in one launch)."""
    from ray_marching_amd import ops
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene
    h, w, steps = 96, 128, 64
    q0 = torch.nn.functional.normalize(torch.tensor([[0.98, 0.05, -0.12, 0.03]]), dim=-1)
    t0 = torch.tensor([[0.15, -0.1, -1.2]])
    wimg = torch.rand(1, h, w, 3, generator=torch.Generator().manual_seed(3)).to(DEV)
    res = {}
    try:
        for cap in (0, capacity):
            ops.bwd_hard_capacity = cap
            scene = make_closed_test_scene()
            loop = H.make_loop(scene, h, w)
            _check_path(scene, kernel_path)
            q, t = q0.to(DEV).requires_grad_(True), t0.to(DEV).requires_grad_(True)
            ops.bwd_tile_cost_sink = torch.zeros(int(__import__("ray_marching_amd")._abi.lib.rm_wave_tiles(1, h, w, 2)),
                                                 dtype=torch.int32, device=DEV)
            (loop(q, t, 0, 1, steps) * wimg).mean().backward()
            res[cap] = ([p.grad.clone() for p in scene.parameters()], q.grad.clone(), t.grad.clone(),
                        ops.bwd_tile_cost_sink.clone())
    finally:
        ops.bwd_hard_capacity = None
        ops.bwd_tile_cost_sink = None
    a, b = res[0], res[capacity]
    assert int(a[3].max()) > 8, "this frame is expected to contain rays that walk many steps"
    if capacity is None:
        assert int(b[3].max()) <= int(a[3].max()) // 2          # the long walks moved to the parallel kernels
    for x, y in zip(a[0] + [a[1], a[2]], b[0] + [b[1], b[2]]):
        assert (x - y).abs().max().item() <= 2e-6 * max(1.0, x.abs().max().item())


@pytest.mark.parametrize("mode", [3, 6, 7])
def test_fused_vjp_of_vignette_tangent_and_spin_shaders(mode, kernel_path):
    """Gradients through the fused frame for the shaders that use the camera pose themselves: vignette
    ((v . col2(q))^3, shader.py:62-66), tangent and spin (brightness x colormap[index], shader.py:107-171; the
    index is piecewise constant, so only the brightness carries a gradient) -- scene parameters, orientation
    and translation vs CPU autograd on the oracle (= the reference's op stream)."""
    h, w, steps = 40, 48, 48
    spec = O.scene_test1_closed()
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    cmap = torch.from_numpy(H.gold("cmap.npz")["cyclic_cmap"])
    q0 = torch.nn.functional.normalize(torch.tensor([[0.98, 0.05, -0.12, 0.03]]), dim=-1)
    t0 = torch.tensor([[0.15, -0.1, -1.2]])
    wimg = torch.rand(1, h, w, 3, generator=torch.Generator().manual_seed(11 + mode), dtype=torch.float64)
    spec_g = O.map_spec(spec, lambda x: x.clone().requires_grad_(True))
    qc, tc = q0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
    (O.render(spec_g, bufs, qc, tc, mode, 2, steps, H.EPS, cmap=cmap) * wimg).mean().backward()
    module = H.spec_to_module(spec)
    loop = H.make_loop(module, h, w)
    loop.shader.cyclic_cmap = cmap.to(DEV)
    _check_path(module, kernel_path)
    qg, tg = q0.to(DEV).requires_grad_(True), t0.to(DEV).requires_grad_(True)
    img = loop(qg, tg, mode, 2, steps)
    assert img.dtype == (torch.float64 if mode in (6, 7) else torch.float32)
    (img * wimg.to(DEV)).mean().backward()
    for name, got, want in (("orientation", qg.grad, qc.grad), ("translation", tg.grad, tc.grad)):
        if want is None:            # the vignette does not depend on the camera position
            assert got is None or float(got.abs().max()) == 0.0
            continue
        scale = max(1e-6, want.abs().max().item())
        err = (got.cpu() - want).abs().max().item()
        print(f"mode {mode} grad {name}: |err|={err:.2e} (scale {scale:.2e})")
        assert err <= 1e-4 * max(1.0, scale) and err <= 5e-3 * scale, (name, err, scale)
    for (pname, want), (_, got) in zip(O.spec_parameters(spec_g), module.named_parameters()):
        if want.grad is None:       # vignette: no scene parameter is involved
            assert got.grad is None or float(got.grad.abs().max()) == 0.0, pname
            continue
        err = (got.grad.cpu() - want.grad).abs().max().item()
        assert err <= 1e-4, (mode, pname, err)


@pytest.mark.parametrize("mode", [5, 1, 2])
@pytest.mark.parametrize("case", range(10))
def test_fused_vjp_of_the_globally_normalised_shaders(case, mode):
    """Mode 5 (shader.py:81-89): lap / max|lap| through clamp and x^(1/2.33): the reference's gradient is finite
    when the pixel of the largest |Laplacian| has a negative one, and NaN in every parameter component that pixel's
    ray reaches when it has a positive one (the clamped value is 0 there and the power has an infinite slope).
    Modes 1 and 2 (shader.py:27-55, normalised by the frame's minimum and maximum): the minimum pixel always is
    x = 0, so the rays of the minimum and the maximum always carry NaN.  Both behaviours are reproduced: NaN patterns
    element for element, finite components against CPU autograd on the oracle -- scene parameters and the pose."""
    h, w, steps = 40, 48, 48
    spec = O.scene_test1_closed() if case % 2 == 0 else O.scene_test2()
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    gen = torch.Generator().manual_seed(300 + case)
    q0 = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0, 0.0, 0.0]]) + 0.1 * torch.randn(1, 4, generator=gen), dim=-1)
    t0 = torch.tensor([[0.0, 0.0, -1.2 if case % 2 == 0 else -3.0]]) + 0.2 * torch.randn(1, 3, generator=gen)
    wimg = torch.rand(1, h, w, 3, generator=gen)
    spec_g = O.map_spec(spec, lambda x: x.clone().requires_grad_(True))
    qc, tc = q0.clone().requires_grad_(True), t0.clone().requires_grad_(True)
    (O.render(spec_g, bufs, qc, tc, mode, 1, steps, H.EPS) * wimg).mean().backward()
    module = H.spec_to_module(spec)
    loop = H.make_loop(module, h, w)
    qg, tg = q0.to(DEV).requires_grad_(True), t0.to(DEV).requires_grad_(True)
    img = loop(qg, tg, mode, 1, steps)
    (img * wimg.to(DEV)).mean().backward()
    wants = [(n, p.grad) for n, p in O.spec_parameters(spec_g)] + [("orientation", qc.grad), ("translation", tc.grad)]
    gots = [p.grad for _, p in module.named_parameters()] + [qg.grad, tg.grad]
    ref_nan = any(wg is not None and torch.isnan(wg).any() for _, wg in wants)
    print(f"mode {mode} case {case}: reference gradient {'NaN' if ref_nan else 'finite'}")
    for (name, want), got in zip(wants, gots):
        if want is None:
            assert got is None or float(got.abs().max()) == 0.0, name
            continue
        got = got.cpu()
        # (a NaN reference gradient leaves exact zeros where no ray reaches a parameter component: the same here)
        assert torch.equal(torch.isnan(want), torch.isnan(got)), (mode, case, name, want, got)
        fin = ~torch.isnan(want)
        if fin.any():
            scale = max(1.0, want[fin].abs().max().item())
            err = (got[fin] - want[fin]).abs().max().item()
            assert err <= 2e-4 * scale, (mode, case, name, err, scale)
