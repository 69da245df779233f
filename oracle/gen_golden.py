"""Generate tests/golden/*.npz from the REFERENCE implementation.  TEST INFRASTRUCTURE.

Run in the build container only (needs /root/reference):

    python oracle/gen_golden.py

Every expected output below is produced by the reference's own modules
(imported through oracle/ref_bridge.py), never by the oracle, so the fixtures
pin both the oracle (tests/test_oracle_golden.py) and the HIP kernels
(tests/test_gpu_*.py) to the reference.  Inputs are seeded or deterministic
pinhole grids; scene parameters are the constants of scene_registry.py.

Fixture families:
  f1_nodes.npz      per-primitive / per-combinator distances on 2048 seeded points
  f2_camera.npz     PinholeCamera buffers + forward for 12x16, two poses
  f3_sphere.npz     config 1: SDFSphere(0.5) 256x256 S=32, 4x subsampled + checksums
  f4_scene2_*.npz   make_test_scene2 frames: p, dist, normals, laplacian, 8 shader modes
  f5_backward.npz   closed make_test_scene, 64x64 S=64: parameter grads (fp32 and fp64)
  f6_ties.npz       subgradient choices at ties
  f7_*fp16*.npz     reference cast to float16 (config 3 numerics) + its own fp16-vs-fp32 spread
  f8_*two_cameras   num_cameras = 2 batch
  f9_many32_*.npz   config-5 scene (32-primitive smooth union in a room), 54x96 S=128, modes 0 and 4
  cmap.npz          the reference's colormap data file (float64 [4096,3])
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import ref_bridge, sdf_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
PX = 3.45e-6
EPS = 5e-2


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def node_specs():
    box = lambda: ("box", {"halfsides": O._t((0.4, 0.7, 1.1))})  # noqa: E731
    return {
        "sphere": O.scene_sphere(0.5),
        "box": box(),
        "plane": ("plane", {}),
        "line": ("line", {"start": O._t((-1.0, 1.0, 2.0)), "end": O._t((1.0, 1.0, 0.0)), "radius": O._t(0.1)}),
        "disk": ("disk", {"radius": O._t(0.8)}),
        "torus": ("torus", {"radius1": O._t(1.0), "radius2": O._t(0.25)}),
        "affine": ("affine", {"translation": O._t((0.1, -0.2, 0.3)),
                              "orientation": O._t((0.9014, 0.25, 0.25, 0.25))}, box()),
        "rounding": ("rounding", {"rounding": O._t(0.07)}, box()),
        "onion": ("onion", {"radius": O._t(0.1)}, O.scene_sphere(1.0)),
        "union": ("union", {}, [O.scene_sphere(0.5), box(), ("torus", {"radius1": O._t(1.0), "radius2": O._t(0.25)})]),
        "smooth_union": ("smooth_union", {"blend_k": O._t(22.0)},
                         [O.scene_sphere(0.5), box(), ("torus", {"radius1": O._t(1.0), "radius2": O._t(0.25)})]),
        "scene1": O.scene_test1(),
        "scene2": O.scene_test2(),
        "scene1_closed": O.scene_test1_closed(),
        "scene_many8": O.scene_many(8),
    }


def gen_f1(ref):
    g = torch.Generator().manual_seed(20260424)
    pts = torch.rand(2048, 3, generator=g) * 6.0 - 3.0
    # a few exact-boundary / degenerate points (origin, faces, axes)
    pts[:8] = torch.tensor([[0, 0, 0], [0.4, 0.7, 1.1], [0.4, 0, 0], [0, 0.7, 0], [1.0, 0, 0],
                            [0, 0, 1.0], [0.5, 0, 0], [-0.4, -0.7, -1.1]], dtype=torch.float32)
    out = {"points": npy(pts)}
    for name, spec in node_specs().items():
        module = ref_bridge.spec_to_reference(ref, spec)
        with torch.no_grad():
            out[name] = npy(module(pts))
            out[name + "_f64"] = npy(module.double()(pts.double()))
    save("f1_nodes.npz", **out)


def gen_f2(ref):
    h, w = 12, 16
    cam = ref.rm.PinholeCamera(1, w, h, PX * h, PX * w, PX * h)
    poses = [
        (torch.tensor([[1.0, 0.0, 0.0, 0.0]]), torch.tensor([[0.0, 0.0, 0.0]])),
        (torch.nn.functional.normalize(torch.tensor([[0.9, 0.1, -0.3, 0.2]]), dim=-1),
         torch.tensor([[0.3, -0.2, -3.0]])),
    ]
    out = {"ray_positions": npy(cam.ray_positions), "ray_directions": npy(cam.ray_directions),
           "hw": np.array([h, w]), "px": np.array(PX)}
    for i, (q, t) in enumerate(poses):
        pos, frames, _, dirs = cam(q, t)
        out.update({f"q{i}": npy(q), f"t{i}": npy(t), f"pos{i}": npy(pos), f"dirs{i}": npy(dirs),
                    f"frames{i}": npy(frames)})
    # multi-camera batch (N=2), the reference supports N cameras
    cam2 = ref.rm.PinholeCamera(2, w, h, PX * h, PX * w, PX * h)
    q2 = torch.cat([poses[0][0], poses[1][0]]); t2 = torch.cat([poses[0][1], poses[1][1]])
    pos, frames, _, dirs = cam2(q2, t2)
    out.update({"q_n2": npy(q2), "t_n2": npy(t2), "pos_n2": npy(pos), "dirs_n2": npy(dirs), "frames_n2": npy(frames)})
    save("f2_camera.npz", **out)


def frame(ref, spec, h, w, q, t, steps, modes, degree=2, dtype=torch.float32):
    scene = ref_bridge.spec_to_reference(ref, spec).to(dtype)
    cam = ref.rm.PinholeCamera(q.shape[0], w, h, PX * h, PX * w, PX * h).to(dtype)
    nrm = ref.rm.SDFNormals(scene, EPS).to(dtype)
    shader_dtype = ref.shader.cyclic_cmap.dtype
    images = {}
    aux = None
    with torch.no_grad():
        for m in modes:
            img, aux = ref_bridge.reference_render(ref, scene, cam, nrm, q.to(dtype), t.to(dtype), m, degree, steps)
            images[m] = img
    assert shader_dtype == torch.float64
    return images, aux


def gen_f3(ref):
    h = w = 256
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.0, 0.0, -2.0]])
    images, aux = frame(ref, O.scene_sphere(0.5), h, w, q, t, 32, [1])
    img, p = images[1], aux["p"]
    finite = torch.isfinite(p).all(-1)
    save("f3_sphere.npz",
         hw=np.array([h, w]), steps=np.array(32), q=npy(q), t=npy(t), stride=np.array(4),
         p_sub=npy(p[:, ::4, ::4]), image_sub=npy(img[:, ::4, ::4, :1]),
         centre_p=npy(p[0, 128, 128]), centre_d=npy(aux["dist"][0, 128, 128]),
         image_mean=np.array(img.double().mean().item()),
         finite_fraction=np.array(finite.double().mean().item()),
         hit_mask_sub=npy(aux["dist"][:, ::4, ::4, 0].abs() < 1e-4))


def gen_f4(ref):
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]])
    q_tilt = torch.nn.functional.normalize(torch.tensor([[0.95, 0.05, 0.25, -0.1]]), dim=-1)
    cases = [
        ("f4_scene2_64_s32_in.npz", 64, 64, q, (0.0, 0.0, 1.0), 32, list(range(8))),
        ("f4_scene2_64_s128_out.npz", 64, 64, q, (0.0, 0.0, -3.0), 128, list(range(8))),
        ("f4_scene2_90x160_s128_tilt.npz", 90, 160, q_tilt, (0.4, -0.3, -3.0), 128, [0, 4]),
    ]
    for name, h, w, quat, tt, steps, modes in cases:
        t = torch.tensor([tt])
        images, aux = frame(ref, O.scene_test2(), h, w, quat, t, steps, modes)
        arrays = {"hw": np.array([h, w]), "steps": np.array(steps), "q": npy(quat), "t": npy(t),
                  "degree": np.array(2), "eps": np.array(EPS), "p": npy(aux["p"])}
        if len(modes) == 8:
            arrays.update({"dist": npy(aux["dist"]), "n": npy(aux["n"]), "lap": npy(aux["lap"])})
        else:
            arrays.update({"n": npy(aux["n"])})
        for m, img in images.items():
            ch = 3 if m in (4, 6, 7) else 1
            arrays[f"mode{m}"] = npy(img[..., :ch]).astype(np.float32 if img.dtype != torch.float64 else np.float64)
        save(name, **arrays)
    # scene1 closed, forward only, for the affine / smooth-union opcodes
    t = torch.tensor([[0.0, 0.0, -1.0]])
    images, aux = frame(ref, O.scene_test1_closed(), 64, 64, q, t, 64, [0, 4])
    save("f4_scene1c_64_s64.npz", hw=np.array([64, 64]), steps=np.array(64), q=npy(q), t=npy(t),
         eps=np.array(EPS), p=npy(aux["p"]), n=npy(aux["n"]), dist=npy(aux["dist"]), lap=npy(aux["lap"]),
         mode0=npy(images[0][..., :1]), mode4=npy(images[4]))


def gen_f5(ref):
    """Config-4 shaped backward: closed scene1, 64x64, S=64; two losses."""
    h = w = 64
    steps = 64
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.0, 0.0, -1.0]])
    out = {"hw": np.array([h, w]), "steps": np.array(steps), "q": npy(q), "t": npy(t), "eps": np.array(EPS)}
    g = torch.Generator().manual_seed(0)
    target = torch.rand(1, h, w, 1, generator=g)
    out["target"] = npy(target)
    for dtype, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for loss_name, mode in (("lambert_mse", 0), ("normal_sq", 4)):
            scene = ref_bridge.spec_to_reference(ref, O.scene_test1_closed()).to(dtype)
            cam = ref.rm.PinholeCamera(1, w, h, PX * h, PX * w, PX * h).to(dtype)
            nrm = ref.rm.SDFNormals(scene, EPS).to(dtype)
            img, aux = ref_bridge.reference_render(ref, scene, cam, nrm, q.to(dtype), t.to(dtype), mode, 1, steps)
            if loss_name == "lambert_mse":
                loss = (img[..., :1] - target.to(dtype)).pow(2).mean()
            else:
                loss = img.pow(2).mean()
            loss.backward()
            out[f"{loss_name}_{tag}_loss"] = np.array(loss.item())
            if tag == "f32":
                out[f"{loss_name}_image"] = npy(img[..., : (1 if mode == 0 else 3)])
            for pname, prm in scene.named_parameters():
                out[f"{loss_name}_{tag}_grad:{pname}"] = npy(prm.grad)
    save("f5_backward.npz", **out)


def gen_f6(ref):
    """Tie / kink subgradients: d/dp of the scene at hand-picked points."""
    out = {}
    cases = {
        # union tie: two identical spheres -> gradient goes to the FIRST child
        "union_tie": (("union", {}, [O.scene_sphere(0.5), O.scene_sphere(0.5)]), [[1.0, 0.0, 0.0]]),
        # box: on a face (q == 0), at the centre (norm of zero vector), on the diagonal (max tie)
        "box_face": (("box", {"halfsides": O._t((0.5, 0.5, 0.5))}), [[0.5, 0.0, 0.0], [0.0, 0.0, 0.0], [0.2, 0.2, 0.2], [1.0, 1.0, 0.1]]),
        # capsule clamp at exactly 0 and 1
        "line_clamp": (("line", {"start": O._t((0.0, 0.0, 0.0)), "end": O._t((1.0, 0.0, 0.0)), "radius": O._t(0.1)}),
                       [[0.0, 0.5, 0.0], [1.0, 0.5, 0.0], [0.5, 0.5, 0.0], [-1.0, 0.5, 0.0]]),
        # onion at d == 0
        "onion_zero": (("onion", {"radius": O._t(0.1)}, O.scene_sphere(1.0)), [[1.0, 0.0, 0.0], [0.5, 0.0, 0.0]]),
        # smooth-union tie splits evenly
        "smooth_tie": (("smooth_union", {"blend_k": O._t(22.0)}, [O.scene_sphere(0.5), O.scene_sphere(0.5)]), [[1.0, 0.0, 0.0]]),
        "disk_edge": (("disk", {"radius": O._t(0.8)}), [[0.3, 0.8, 0.0], [0.3, 0.2, 0.1], [0.0, 1.5, 0.0]]),
    }
    for name, (spec, pts) in cases.items():
        spec = O.map_spec(spec, lambda x: x.clone())
        module = ref_bridge.spec_to_reference(ref, spec)
        p = torch.tensor(pts, dtype=torch.float32, requires_grad=True)
        d = module(p)
        d.sum().backward()
        out[name + "_points"] = npy(p)
        out[name + "_d"] = npy(d)
        out[name + "_grad_p"] = npy(p.grad)
        for pname, prm in module.named_parameters():
            out[f"{name}_grad:{pname}"] = npy(prm.grad)
    save("f6_ties.npz", **out)


def gen_f7(ref):
    """Config-3 style fp16 frame: the reference module cast with .to(float16) (every ATen op rounds
    to fp16).  Also records the reference's own fp16-vs-fp32 spread, which sets the tolerance."""
    h, w, steps = 90, 160, 32
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.0, 0.0, -3.0]])
    img16, aux16 = frame(ref, O.scene_test2(), h, w, q, t, steps, [0, 4], dtype=torch.float16)
    img32, aux32 = frame(ref, O.scene_test2(), h, w, q, t, steps, [0, 4], dtype=torch.float32)
    save("f7_scene2_fp16_90x160_s32.npz", hw=np.array([h, w]), steps=np.array(steps), q=npy(q), t=npy(t),
         eps=np.array(EPS), p16=npy(aux16["p"].float()), n16=npy(aux16["n"].float()),
         mode0_16=npy(img16[0][..., :1].float()), mode4_16=npy(img16[4].float()),
         mode0_32=npy(img32[0][..., :1]), mode4_32=npy(img32[4]), p32=npy(aux32["p"]),
         ref_spread_p=np.array((aux16["p"].float() - aux32["p"]).abs().max().item()),
         ref_spread_mode4=np.array((img16[4].float() - img32[4]).abs().max().item()),
         ref_spread_mode0=np.array((img16[0].float() - img32[0]).abs().max().item()))


def gen_f8(ref):
    """Two cameras in one batch (num_cameras = 2): the reference renders [2,H,W,3]."""
    h, w, steps = 40, 56, 48
    q = torch.cat([torch.tensor([[1.0, 0.0, 0.0, 0.0]]),
                   torch.nn.functional.normalize(torch.tensor([[0.95, 0.05, 0.25, -0.1]]), dim=-1)])
    t = torch.tensor([[0.0, 0.0, -3.0], [0.4, -0.3, -3.0]])
    images, aux = frame(ref, O.scene_test2(), h, w, q, t, steps, [0, 1, 4])
    save("f8_scene2_two_cameras.npz", hw=np.array([h, w]), steps=np.array(steps), q=npy(q), t=npy(t),
         eps=np.array(EPS), p=npy(aux["p"]), n=npy(aux["n"]), mode0=npy(images[0][..., :1]),
         mode1=npy(images[1][..., :1]), mode4=npy(images[4]))


def gen_f9(ref):
    """Config-5 scene at a size the reference renders in seconds.  The smooth union goes through
    torch.logsumexp = MKL VML exp/log, whose bits depend on the host CPU: this fixture holds the build
    container's (Intel) result, and the GPU test also reports how far the GPU box's own CPU lands from it."""
    h, w, steps = 54, 96, 128
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]]); t = torch.tensor([[0.0, 0.0, -4.5]])
    images, aux = frame(ref, O.scene_many(32), h, w, q, t, steps, [0, 4], degree=1)
    # the reference's own float64 render of the same frame (same fp32 parameter / camera values, cast like .to(float64)
    # casts them): |fp32 - fp64| per pixel is the reference's own conditioning there, which the GPU test uses to say
    # WHICH pixels may differ by more than 1e-5 (grazing rays at blend creases)
    images64, aux64 = frame(ref, O.scene_many(32), h, w, q, t, steps, [0, 4], degree=1, dtype=torch.float64)
    save("f9_many32_54x96_s128.npz", hw=np.array([h, w]), steps=np.array(steps), q=npy(q), t=npy(t),
         eps=np.array(EPS), p=npy(aux["p"]), n=npy(aux["n"]), dist=npy(aux["dist"]),
         mode0=npy(images[0][..., :1]), mode4=npy(images[4]),
         p_f64=npy(aux64["p"]), mode0_f64=npy(images64[0][..., :1]), mode4_f64=npy(images64[4]))


def main():
    if len(sys.argv) > 1:          # python oracle/gen_golden.py f9  -> only that family
        torch.set_num_threads(8)
        ref = ref_bridge.load_reference()
        for name in sys.argv[1:]:
            globals()["gen_" + name](ref)
        return
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref = ref_bridge.load_reference()
    gen_f1(ref)
    gen_f2(ref)
    gen_f3(ref)
    gen_f4(ref)
    gen_f5(ref)
    gen_f6(ref)
    gen_f7(ref)
    gen_f8(ref)
    gen_f9(ref)
    save("cmap.npz", cyclic_cmap=npy(ref.shader.cyclic_cmap))


if __name__ == "__main__":
    main()
