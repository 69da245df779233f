"""Pins the oracle to the imported reference (build container only).

Skipped wherever /root/reference is absent (e.g. the GPU box); there the oracle
is pinned by tests/test_oracle_golden.py against fixtures generated from the
same reference by oracle/gen_golden.py.
"""
import pytest
import torch

from oracle import ref_bridge, sdf_oracle as O

pytestmark = pytest.mark.skipif(not ref_bridge.reference_available(),
                                reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    return ref_bridge.load_reference()


def _same(a, b):
    """Bitwise equality that also requires NaNs in the same places (open scenes
    produce NaN normals for rays that miss)."""
    return torch.equal(torch.nan_to_num(a, nan=-7.0, posinf=3e38, neginf=-3e38),
                       torch.nan_to_num(b, nan=-7.0, posinf=3e38, neginf=-3e38))


def _points(n=4096, seed=0, lo=-3.0, hi=3.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, generator=g) * (hi - lo) + lo


NODE_SPECS = {
    "sphere": lambda: O.scene_sphere(0.5),
    "box": lambda: ("box", {"halfsides": O._t((0.4, 0.7, 1.1))}),
    "plane": lambda: ("plane", {}),
    "line": lambda: ("line", {"start": O._t((-1.0, 1.0, 2.0)), "end": O._t((1.0, 1.0, 0.0)),
                              "radius": O._t(0.1)}),
    "disk": lambda: ("disk", {"radius": O._t(0.8)}),
    "torus": lambda: ("torus", {"radius1": O._t(1.0), "radius2": O._t(0.25)}),
    "affine": lambda: ("affine", {"translation": O._t((0.1, -0.2, 0.3)),
                                  "orientation": O._t((0.9014, 0.25, 0.25, 0.25))},
                       ("box", {"halfsides": O._t((0.4, 0.7, 1.1))})),
    "rounding": lambda: ("rounding", {"rounding": O._t(0.07)}, ("box", {"halfsides": O._t((0.4, 0.7, 1.1))})),
    "onion": lambda: ("onion", {"radius": O._t(0.1)}, O.scene_sphere(1.0)),
    "scene1": O.scene_test1,
    "scene2": O.scene_test2,
    "scene1_closed": O.scene_test1_closed,
    "scene_many": lambda: O.scene_many(8),
}


@pytest.mark.parametrize("name", sorted(NODE_SPECS))
def test_sdf_nodes_bitwise(ref, name):
    spec = NODE_SPECS[name]()
    module = ref_bridge.spec_to_reference(ref, spec)
    pts = _points()
    with torch.no_grad():
        want = module(pts)
        got = O.sdf_eval(spec, pts)
    assert got.shape == want.shape == (pts.shape[0], 1)
    assert torch.equal(got, want)


def test_registry_scenes_match_factories(ref):
    pts = _points(seed=3)
    with torch.no_grad():
        assert torch.equal(ref.registry.make_test_scene()(pts), O.sdf_eval(O.scene_test1(), pts))
        assert torch.equal(ref.registry.make_test_scene2()(pts), O.sdf_eval(O.scene_test2(), pts))
    names_ref = [n for n, _ in ref.registry.make_test_scene().named_parameters()]
    assert names_ref == [n for n, _ in O.spec_parameters(O.scene_test1())]
    names_ref = [n for n, _ in ref.registry.make_test_scene2().named_parameters()]
    assert names_ref == [n for n, _ in O.spec_parameters(O.scene_test2())]


def test_quaternion_helpers(ref):
    g = torch.Generator().manual_seed(1)
    u, v = torch.randn(512, 3, generator=g), torch.randn(512, 3, generator=g)
    p, q = torch.randn(512, 4, generator=g), torch.randn(512, 4, generator=g)
    assert torch.equal(O.cross(u, v), ref.Q.cross_product(u, v))
    assert torch.equal(O.quat_rotate(u, q), ref.Q.rotation(u, q))
    assert torch.equal(O.quat_conj(q), ref.Q.conjugate(q))
    assert torch.equal(O.quat_multiply(p, q), ref.Q.multiply(p, q))
    assert torch.equal(O.quat_to_so3(q), ref.Q.QuaternionToSO3()(q))


@pytest.mark.parametrize("hw", [(12, 16), (64, 64)])
def test_camera(ref, hw):
    h, w = hw
    px = 3.45e-6
    cam = ref.rm.PinholeCamera(1, w, h, px * h, px * w, px * h)
    origins, directions = O.camera_buffers(1, w, h, px * h, px * w, px * h)
    assert torch.equal(origins, cam.ray_positions)
    assert torch.equal(directions, cam.ray_directions)
    q = torch.nn.functional.normalize(torch.tensor([[0.9, 0.1, -0.3, 0.2]]), dim=-1)
    t = torch.tensor([[0.3, -0.2, -3.0]])
    pos_r, frames_r, _, dirs_r = cam(q, t)
    pos, frames, dirs = O.camera_forward(origins, directions, q, t)
    assert torch.equal(pos, pos_r) and torch.equal(dirs, dirs_r) and torch.equal(frames, frames_r)


@pytest.mark.parametrize("scene_name,cam_t,steps", [
    ("scene2", (0.0, 0.0, 1.0), 32), ("scene2", (0.0, 0.0, -3.0), 64),
    ("scene1_closed", (0.0, 0.0, -1.0), 32), ("sphere", (0.0, 0.0, -2.0), 32),
])
def test_full_frame_all_modes(ref, scene_name, cam_t, steps):
    h, w, px, eps = 24, 32, 3.45e-6, 5e-2
    spec = NODE_SPECS[scene_name]()
    scene = ref_bridge.spec_to_reference(ref, spec)
    cam = ref.rm.PinholeCamera(1, w, h, px * h, px * w, px * h)
    nrm = ref.rm.SDFNormals(scene, eps)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]])
    t = torch.tensor([cam_t])
    bufs = O.camera_buffers(1, w, h, px * h, px * w, px * h)
    for mode in range(8):
        with torch.no_grad():
            want, aux_r = ref_bridge.reference_render(ref, scene, cam, nrm, q, t, mode, 2, steps)
            got, aux = O.render(spec, bufs, q, t, mode, 2, steps, eps, cmap=ref.shader.cyclic_cmap,
                                return_aux=True)
        for key in ("p", "dist", "n", "lap"):
            assert _same(aux[key], aux_r[key]), (mode, key)
        assert got.shape == want.shape == (1, h, w, 3)
        # open-scene rays can be NaN; compare NaN-aware.
        assert _same(got, want), mode


def test_backward_matches_reference_autograd(ref):
    h, w, px, eps, steps = 16, 16, 3.45e-6, 5e-2, 24
    spec = O.map_spec(O.scene_test1_closed(), lambda x: x.clone().requires_grad_(True))
    scene = ref_bridge.spec_to_reference(ref, spec)
    cam = ref.rm.PinholeCamera(1, w, h, px * h, px * w, px * h)
    nrm = ref.rm.SDFNormals(scene, eps)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]])
    t = torch.tensor([[0.0, 0.0, -1.0]])
    img_r, _ = ref_bridge.reference_render(ref, scene, cam, nrm, q, t, 0, 1, steps)
    img_r.pow(2).mean().backward()
    bufs = O.camera_buffers(1, w, h, px * h, px * w, px * h)
    img = O.render(spec, bufs, q, t, 0, 1, steps, eps)
    img.pow(2).mean().backward()
    ref_grads = dict(scene.named_parameters())
    for name, tensor in O.spec_parameters(spec):
        # same maths, but autograd accumulates the per-pixel contributions in a
        # different order (stack/unbind here vs index gathers there): not bitwise.
        torch.testing.assert_close(tensor.grad, ref_grads[name].grad, rtol=1e-4, atol=2e-7, msg=name)
