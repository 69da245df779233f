"""Import the reference implementation (build container only).  TEST INFRASTRUCTURE.

``/root/reference`` exists only in the build container; it never travels to the
GPU box.  This bridge is used by ``oracle/gen_golden.py`` (fixture generation)
and by ``tests/test_oracle_vs_reference.py`` (skipped when the tree is absent).

The reference resolves ``./data/cyclic_cmap.pt`` relative to the current
directory, both at ``Shader()`` construction (rendering/shader.py:177) and at
import time of ``rendering.shader`` (:269), so the import happens with cwd set
to the reference root.  ``control.py`` needs pynput/pyautogui (absent), so
``RenderLoop.forward`` (control.py:239-258) is re-composed from its importable
parts in ``reference_render`` below.
"""
from __future__ import annotations

import contextlib
import io
import os
import sys
from types import SimpleNamespace

REFERENCE_ROOT = os.environ.get("RM_REFERENCE_ROOT", "/root/reference")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "rendering", "ray_marching.py"))


_cached = None


def load_reference():
    """Returns a namespace with the reference's hot-path modules."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    sys.dont_write_bytecode = True
    old_cwd = os.getcwd()
    os.chdir(REFERENCE_ROOT)
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            import quaternion as ref_q  # noqa: F401
            import rendering.ray_marching as ref_rm
            import rendering.shader as ref_shader
            import scene.primitives as ref_prims
            import scene.transformations as ref_tf
            import scene.scene_registry as ref_registry
            shader = ref_shader.Shader()
    finally:
        os.chdir(old_cwd)
        sys.path.remove(REFERENCE_ROOT)
    _cached = SimpleNamespace(Q=ref_q, rm=ref_rm, shader_mod=ref_shader, prims=ref_prims,
                              tf=ref_tf, registry=ref_registry, shader=shader)
    return _cached


def module_to_spec(module):
    """Reference ``nn.Module`` scene tree -> oracle scene spec (parameters are
    shared, not copied, so autograd through the spec reaches the module)."""
    name = type(module).__name__
    if name == "SDFSphere":
        return ("sphere", {"radius": module.radius})
    if name == "SDFBox":
        return ("box", {"halfsides": module.halfsides})
    if name == "SDFPlane":
        return ("plane", {})
    if name == "SDFLine":
        return ("line", {"start": module.start, "end": module.end, "radius": module.radius})
    if name == "SDFDisk":
        return ("disk", {"radius": module.radius})
    if name == "SDFTorus":
        return ("torus", {"radius1": module.radius1, "radius2": module.radius2})
    if name == "SDFAffineTransformation":
        return ("affine", {"translation": module.translation, "orientation": module.orientation},
                module_to_spec(module.sdf))
    if name == "SDFSmoothUnion":
        return ("smooth_union", {"blend_k": module.blend_k}, [module_to_spec(m) for m in module.sdfs])
    if name == "SDFUnion":
        return ("union", {}, [module_to_spec(m) for m in module.sdfs])
    if name == "SDFRounding":
        return ("rounding", {"rounding": module.rounding}, module_to_spec(module.sdf))
    if name == "SDFOnion":
        return ("onion", {"radius": module.radius}, module_to_spec(module.sdf))
    raise TypeError(f"not a reference SDF module: {name}")


def spec_to_reference(ref, spec):
    """Oracle scene spec -> reference ``nn.Module`` tree (values copied)."""
    kind, prm = spec[0], spec[1]
    P, T = ref.prims, ref.tf

    def f(x):
        return x.detach().tolist()

    if kind == "sphere":
        return P.SDFSphere(f(prm["radius"]))
    if kind == "box":
        return P.SDFBox(tuple(f(prm["halfsides"])))
    if kind == "plane":
        return P.SDFPlane()
    if kind == "line":
        return P.SDFLine(tuple(f(prm["start"])), tuple(f(prm["end"])), f(prm["radius"]))
    if kind == "disk":
        return P.SDFDisk(f(prm["radius"]))
    if kind == "torus":
        return P.SDFTorus(f(prm["radius1"]), f(prm["radius2"]))
    if kind == "affine":
        return T.SDFAffineTransformation(spec_to_reference(ref, spec[2]),
                                         orientation=f(prm["orientation"]),
                                         translation=f(prm["translation"]))
    if kind == "smooth_union":
        return T.SDFSmoothUnion([spec_to_reference(ref, c) for c in spec[2]], f(prm["blend_k"]))
    if kind == "union":
        return T.SDFUnion([spec_to_reference(ref, c) for c in spec[2]])
    if kind == "rounding":
        return T.SDFRounding(spec_to_reference(ref, spec[2]), f(prm["rounding"]))
    if kind == "onion":
        return T.SDFOnion(spec_to_reference(ref, spec[2]), f(prm["radius"]))
    raise ValueError(kind)


def reference_render(ref, scene, camera, normals_mod, orientations, translations,
                     mode, degree, steps):
    """The body of RenderLoop.forward (control.py:239-258) composed from the
    reference's own camera / marcher / normals / shader modules."""
    pixel_pos, frames, ray_pos, ray_dirs = camera(orientation=orientations, translation=translations)
    marched = ref.rm.SDFMarcher(scene)(ray_pos, ray_dirs, steps)
    dist = scene(marched)
    n, lap = normals_mod(marched)
    img = ref.shader(pixel_pos, orientations, frames, ray_dirs, marched, n, lap, dist,
                     mode=mode, degree=degree)
    h, w = ray_pos.shape[1], ray_pos.shape[2]
    aux = {"pos": ray_pos, "dirs": ray_dirs, "p": marched, "dist": dist, "n": n, "lap": lap,
           "frames": frames}
    return img.expand(-1, h, w, 3), aux
