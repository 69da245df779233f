"""Oracle vs the committed reference outputs (tests/golden, made by oracle/gen_golden.py).

Runs wherever the tests run (no reference tree needed).  Forward values are
bit-exact on the build container's torch; a tiny tolerance is allowed so a
different host CPU / ATen vectorisation on the GPU box does not turn rounding
into a failure.
"""
import os

import numpy as np
import pytest
import torch

from oracle import sdf_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
PX = 3.45e-6


def load(name):
    return np.load(os.path.join(GOLD, name))


def close(a, b, atol=2e-6, rtol=1e-5):
    a = torch.as_tensor(a)
    b = torch.as_tensor(b)
    nan_a, nan_b = torch.isnan(a), torch.isnan(b)
    assert torch.equal(nan_a, nan_b)
    torch.testing.assert_close(torch.nan_to_num(a), torch.nan_to_num(b), atol=atol, rtol=rtol)


def node_spec(name):
    from oracle.gen_golden import node_specs  # spec constants only; does not touch the reference
    return node_specs()[name]


NODES = ["sphere", "box", "plane", "line", "disk", "torus", "affine", "rounding", "onion", "union",
         "smooth_union", "scene1", "scene2", "scene1_closed", "scene_many8"]


@pytest.mark.parametrize("name", NODES)
def test_nodes(name):
    g = load("f1_nodes.npz")
    pts = torch.from_numpy(g["points"])
    spec = node_spec(name)
    with torch.no_grad():
        close(O.sdf_eval(spec, pts), g[name])
        spec64 = O.map_spec(spec, lambda x: x.double())
        close(O.sdf_eval(spec64, pts.double()), g[name + "_f64"], atol=1e-12, rtol=1e-10)


def test_camera():
    g = load("f2_camera.npz")
    h, w = (int(x) for x in g["hw"])
    origins, directions = O.camera_buffers(1, w, h, PX * h, PX * w, PX * h)
    close(origins, g["ray_positions"], atol=1e-12)
    close(directions, g["ray_directions"], atol=1e-7)
    for i in (0, 1):
        pos, frames, dirs = O.camera_forward(origins, directions, torch.from_numpy(g[f"q{i}"]),
                                             torch.from_numpy(g[f"t{i}"]))
        close(pos, g[f"pos{i}"]); close(dirs, g[f"dirs{i}"]); close(frames, g[f"frames{i}"])
    o2, d2 = O.camera_buffers(2, w, h, PX * h, PX * w, PX * h)
    pos, frames, dirs = O.camera_forward(o2, d2, torch.from_numpy(g["q_n2"]), torch.from_numpy(g["t_n2"]))
    close(pos, g["pos_n2"]); close(dirs, g["dirs_n2"]); close(frames, g["frames_n2"])


def test_sphere_config1():
    g = load("f3_sphere.npz")
    h, w = (int(x) for x in g["hw"])
    bufs = O.camera_buffers(1, w, h, PX * h, PX * w, PX * h)
    with torch.no_grad():
        img, aux = O.render(O.scene_sphere(0.5), bufs, torch.from_numpy(g["q"]), torch.from_numpy(g["t"]),
                            1, 1, int(g["steps"]), 5e-2, return_aux=True)
    s = int(g["stride"])
    close(aux["p"][:, ::s, ::s], g["p_sub"], atol=1e-3, rtol=1e-5)  # missed rays sit at |p| ~ 1e8
    close(img[:, ::s, ::s, :1], g["image_sub"])
    # analytic known answers: centre ray hits the sphere of radius 0.5 head-on
    assert abs(float(aux["p"][0, 128, 128].norm()) - 0.5) < 1e-6
    assert abs(img.double().mean().item() - float(g["image_mean"])) < 1e-7
    assert abs(float(g["image_mean"]) - 0.75755148) < 1e-6  # SURVEY 8c anchor


@pytest.mark.parametrize("name", ["f4_scene2_64_s32_in.npz", "f4_scene2_64_s128_out.npz",
                                  "f4_scene2_90x160_s128_tilt.npz", "f4_scene1c_64_s64.npz"])
def test_frames(name):
    g = load(name)
    h, w = (int(x) for x in g["hw"])
    spec = O.scene_test1_closed() if "scene1c" in name else O.scene_test2()
    bufs = O.camera_buffers(1, w, h, PX * h, PX * w, PX * h)
    cmap = torch.from_numpy(load("cmap.npz")["cyclic_cmap"])
    degree = int(g["degree"]) if "degree" in g.files else 1
    modes = [int(k[4:]) for k in g.files if k.startswith("mode")]
    for m in modes:
        with torch.no_grad():
            img, aux = O.render(spec, bufs, torch.from_numpy(g["q"]), torch.from_numpy(g["t"]), m, degree,
                                int(g["steps"]), float(g["eps"]), cmap=cmap, return_aux=True)
        want = g[f"mode{m}"]
        close(img[..., : want.shape[-1]], want)
    close(aux["p"], g["p"]); close(aux["n"], g["n"])
    for k in ("dist", "lap"):
        if k in g.files:
            close(aux[k], g[k], atol=1e-4 if k == "lap" else 2e-6)


@pytest.mark.parametrize("loss_name,mode", [("lambert_mse", 0), ("normal_sq", 4)])
def test_backward(loss_name, mode):
    g = load("f5_backward.npz")
    h, w = (int(x) for x in g["hw"])
    for dtype, tag, rtol, atol in ((torch.float32, "f32", 2e-4, 1e-7), (torch.float64, "f64", 1e-6, 1e-12)):
        # the reference builds fp32 parameters and casts (.to(dtype)); do the same
        spec = O.map_spec(O.scene_test1_closed(), lambda x: x.to(dtype).requires_grad_(True))
        bufs = tuple(b.to(dtype) for b in O.camera_buffers(1, w, h, PX * h, PX * w, PX * h))
        img = O.render(spec, bufs, torch.from_numpy(g["q"]).to(dtype), torch.from_numpy(g["t"]).to(dtype),
                       mode, 1, int(g["steps"]), float(g["eps"]))
        if loss_name == "lambert_mse":
            loss = (img[..., :1] - torch.from_numpy(g["target"]).to(dtype)).pow(2).mean()
        else:
            loss = img.pow(2).mean()
        loss.backward()
        assert abs(loss.item() - float(g[f"{loss_name}_{tag}_loss"])) < 1e-6
        for pname, prm in O.spec_parameters(spec):
            want = torch.from_numpy(g[f"{loss_name}_{tag}_grad:{pname}"])
            torch.testing.assert_close(prm.grad, want, rtol=rtol, atol=atol, msg=f"{tag} {pname}")


def test_tie_subgradients():
    g = load("f6_ties.npz")
    from oracle.gen_golden import gen_f6  # noqa: F401  (documented source of the cases)
    cases = {
        "union_tie": ("union", {}, [O.scene_sphere(0.5), O.scene_sphere(0.5)]),
        "box_face": ("box", {"halfsides": O._t((0.5, 0.5, 0.5))}),
        "line_clamp": ("line", {"start": O._t((0.0, 0.0, 0.0)), "end": O._t((1.0, 0.0, 0.0)), "radius": O._t(0.1)}),
        "onion_zero": ("onion", {"radius": O._t(0.1)}, O.scene_sphere(1.0)),
        "smooth_tie": ("smooth_union", {"blend_k": O._t(22.0)}, [O.scene_sphere(0.5), O.scene_sphere(0.5)]),
        "disk_edge": ("disk", {"radius": O._t(0.8)}),
    }
    for name, spec in cases.items():
        spec = O.map_spec(spec, lambda x: x.clone().requires_grad_(True))
        p = torch.from_numpy(g[name + "_points"]).requires_grad_(True)
        d = O.sdf_eval(spec, p)
        d.sum().backward()
        close(d, g[name + "_d"])
        close(p.grad, g[name + "_grad_p"])
        for pname, prm in O.spec_parameters(spec):
            close(prm.grad, g[f"{name}_grad:{pname}"])


def test_known_answers():
    """Analytic pins that do not depend on any fixture."""
    p = torch.tensor([[2.0, 0.0, 0.0], [0.0, 0.0, 0.0], [0.0, 3.0, 4.0]])
    assert O.sdf_eval(O.scene_sphere(0.5), p).flatten().tolist() == [1.5, -0.5, 4.5]
    box = ("box", {"halfsides": O._t((1.0, 2.0, 3.0))})
    assert O.sdf_eval(box, torch.tensor([[2.0, 0, 0], [0, 0, 0], [4.0, 6.0, 3.0]])).flatten().tolist() == [1.0, -1.0, 5.0]
    torus = ("torus", {"radius1": O._t(1.0), "radius2": O._t(0.25)})
    assert O.sdf_eval(torus, torch.tensor([[1.0, 0, 0], [0, 0, 2.0], [0.0, 1.0, 0.0]])).flatten().tolist() == \
        pytest.approx([-0.25, 0.75, 2 ** 0.5 - 0.25], abs=1e-7)
    line = ("line", {"start": O._t((0.0, 0, 0)), "end": O._t((2.0, 0, 0)), "radius": O._t(0.5)})
    assert O.sdf_eval(line, torch.tensor([[1.0, 1.0, 0], [-1.0, 0, 0], [3.0, 0.0, 0.0]])).flatten().tolist() == \
        pytest.approx([0.5, 0.5, 0.5], abs=1e-7)
    # tetrahedral normal of a plane is exact; laplacian of a sphere SDF is 2/r
    n, lap = O.normals(("plane", {}), torch.tensor([[0.3, 0.2, -0.4]]), 5e-2)
    assert n.flatten().tolist() == pytest.approx([1.0, 0.0, 0.0], abs=1e-6)
    assert abs(lap.item()) < 1e-3
    n, lap = O.normals(O.map_spec(O.scene_sphere(0.5), lambda x: x.double()),
                       torch.tensor([[0.0, 0.0, 2.0]], dtype=torch.float64), 1e-3)
    assert n.flatten().tolist() == pytest.approx([0.0, 0.0, 1.0], abs=1e-9)
    assert lap.item() == pytest.approx(-2.0 / 2.0, rel=1e-3)  # sign: f(p) - mean(taps) = -(eps^2/6) * lap f


def test_two_camera_batch():
    g = load("f8_scene2_two_cameras.npz")
    h, w = (int(x) for x in g["hw"])
    bufs = O.camera_buffers(2, w, h, PX * h, PX * w, PX * h)
    for m in (0, 1, 4):
        with torch.no_grad():
            img, aux = O.render(O.scene_test2(), bufs, torch.from_numpy(g["q"]), torch.from_numpy(g["t"]), m, 2,
                                int(g["steps"]), float(g["eps"]), return_aux=True)
        want = g[f"mode{m}"]
        assert img.shape == (2, h, w, 3)
        close(img[..., : want.shape[-1]], want)
    close(aux["p"], g["p"]); close(aux["n"], g["n"])


def test_fp16_reference_path():
    """The oracle run in float16 reproduces the reference cast with .to(float16) (config 3)."""
    g = load("f7_scene2_fp16_90x160_s32.npz")
    h, w = (int(x) for x in g["hw"])
    spec = O.map_spec(O.scene_test2(), lambda x: x.half())
    bufs = tuple(b.half() for b in O.camera_buffers(1, w, h, PX * h, PX * w, PX * h))
    with torch.no_grad():
        img, aux = O.render(spec, bufs, torch.from_numpy(g["q"]).half(), torch.from_numpy(g["t"]).half(), 4, 2,
                            int(g["steps"]), float(g["eps"]), return_aux=True)
    assert img.dtype == torch.float16
    d = (aux["p"].float() - torch.from_numpy(g["p16"])).abs()
    assert d.max().item() <= 1e-2 and (d > 0).float().mean().item() < 0.01   # bit-exact up to host ATen differences
    # the reference's own fp16-vs-fp32 spread is large: fp16 parity can only be statistical
    assert float(g["ref_spread_p"]) > 0.1


def test_config5_scene_fixture_and_its_float64_render():
    """f9 (the 32-primitive smooth union of config 5, 54x96x128): the oracle in float64 reproduces the reference's own
    float64 render held in the fixture, and the float32 oracle is within 1e-5 of the float32 fixture on every pixel
    the reference resolves to 1e-5 -- i.e. where its float32 and float64 renders agree that closely (on the build
    container's MKL it is bit-exact everywhere; the grazing-ray pixels are where another host's exp ulps may show)."""
    g = load("f9_many32_54x96_s128.npz")
    h, w = (int(x) for x in g["hw"]); steps = int(g["steps"])
    q, t = torch.from_numpy(g["q"]), torch.from_numpy(g["t"])
    bufs = O.camera_buffers(1, w, h, PX * h, PX * w, PX * h)
    spec64 = O.map_spec(O.scene_many(32), lambda x: x.double())
    bufs64 = tuple(b.double() for b in bufs)
    tetra64 = O.tetra_constants(float(g["eps"]), torch.float64)
    for mode in (0, 4):
        with torch.no_grad():
            img64, aux64 = O.render(spec64, bufs64, q.double(), t.double(), mode, 1, steps, float(g["eps"]),
                                    return_aux=True, tetra=tetra64)
            img32 = O.render(O.scene_many(32), bufs, q, t, mode, 1, steps, float(g["eps"]))
        want64, want32 = g[f"mode{mode}_f64"], g[f"mode{mode}"]
        ch = want32.shape[-1]
        assert img64.dtype == torch.float64
        close(img64[..., :ch], want64, atol=1e-9, rtol=1e-9)
        err = np.abs(img32[..., :ch].double().numpy() - want32.astype(np.float64))
        spread = np.abs(want32.astype(np.float64) - want64).max(axis=-1, keepdims=True)
        assert not ((err > 1e-5) & (spread <= 1e-5)).any()
        assert 0 < (spread > 1e-5).sum() < 0.05 * spread.size          # the ill-conditioned pixels exist and are few
    close(aux64["p"], g["p_f64"], atol=1e-9, rtol=1e-9)
