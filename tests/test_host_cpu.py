"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol of
include/rm_abi.h, argument validation returns error codes (nothing is launched), the scene
compiler lowers the reference's scenes as documented, and the module surface mirrors the
reference (names, buffers, parameter order)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from oracle import sdf_oracle as O
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rm_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ray_marching_amd import _abi
    names = declared_functions()
    assert len(names) >= 18
    lib = C.CDLL(_abi.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/rm_abi.h but not exported"
    assert set(names) == set(_abi.EXPORTED_SYMBOLS), set(names) ^ set(_abi.EXPORTED_SYMBOLS)
    assert lib.rm_abi_version() == _abi.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "rm_abi.h")).read()
    assert f"#define RM_ABI_VERSION {_abi.ABI_VERSION}" in header
    assert f"#define RM_WORK_WORDS" in header and _abi.WORK_WORDS == 64 + 64 * 32 + 8 * 4 * 32 + 64 * 32


def test_bad_arguments_return_error_codes_without_launching():
    from ray_marching_amd import _abi
    lib = _abi.lib
    assert lib.rm_sdf_forward(None, None, None, 4, 0, None) == -1
    assert b"scene" in lib.rm_last_error()
    bogus = _abi.RmScene(program=1, params=1, n_instr=0, n_params=0, n_derived=0, stack_floats=0, n_slots=0)
    assert lib.rm_sdf_forward(bogus, None, None, 4, 0, None) == -1
    ok = _abi.RmScene(program=16, params=16, n_instr=1, n_params=1, n_derived=0, stack_floats=0, n_slots=0)
    assert lib.rm_sdf_forward(ok, None, None, 4, 0, None) == -1          # null buffers
    assert lib.rm_sdf_forward(ok, 16, 16, 4, 2, None) == -1              # fp64 points: not an I/O type
    assert b"dtype" in lib.rm_last_error()
    assert lib.rm_sdf_forward(ok, None, None, 0, 0, None) == 0           # empty input: nothing to do
    refs_only = _abi.RmScene(program=16, params=None, param_refs=16, n_instr=1, n_params=1, n_derived=0, stack_floats=0, n_slots=0)
    assert lib.rm_sdf_forward(refs_only, None, None, 0, 1, None) == 0    # a gather table instead of the packed block
    neither = _abi.RmScene(program=16, params=None, param_refs=None, n_instr=1, n_params=1, n_derived=0, stack_floats=0, n_slots=0)
    assert lib.rm_sdf_forward(neither, None, None, 0, 0, None) == -1
    assert lib.rm_shade_finish(None, None, 0, 10, None, 1, 0, None) == -1
    assert lib.rm_shade_finish(16, 16, 1, 10, 16, 1, 0, None) == -1         # fp16 image cannot be normalised in place
    assert lib.rm_minmax_init(None, None) == -1
    cam = _abi.RmCamera(ray_positions=16, ray_directions=16, num_cameras=1, height=4, width=4, dtype=0)
    tet = _abi.RmTetra()

    def render(mode, r0=0, r1=4, image_dtype=0, cam_=cam, cmap=None, cmap_dtype=0):
        return lib.rm_render_forward(ok, cam_, tet, 16, 16, 16, image_dtype, None, None, None, None, None, None, cmap,
                                     0 if cmap is None else 8, cmap_dtype, mode, 1, 8, r0, r1, 0, None, None, None, 0, None)

    assert render(9) == -1 and b"mode" in lib.rm_last_error()
    assert render(0, 2, 9) == -1                                          # band outside the frame
    assert render(1) == -1 and b"minmax" in lib.rm_last_error()
    assert render(0, image_dtype=2) == -1 and b"F64" in lib.rm_last_error()   # float64 image only for modes 6, 7
    assert render(6) == -1 and b"colormap" in lib.rm_last_error()
    assert render(6, cmap=16, cmap_dtype=7) == -1

    def regen(steps=8, p_final=16, minmax=16, flags=1 | 2 | 4 | 8, traj=None):
        return lib.rm_render_forward(ok, cam, tet, 16, 16, 16, 0, None, p_final, traj, None, None, minmax, None, 0, 0, 0, 1, steps,
                                     0, 4, flags, None, None, None, 0, None)

    # ray regeneration states its requirements instead of doing something else
    for bad in (dict(p_final=None), dict(minmax=None), dict(steps=6), dict(flags=2 | 4 | 8), dict(flags=1 | 4 | 8), dict(traj=16)):
        assert regen(**bad) == -1 and b"RM_FLAG_REGEN" in lib.rm_last_error(), bad
    assert lib.rm_tile_order_from_cost(None, 10, 8, None, None, None) == -1
    assert lib.rm_tile_order_from_cost(16, 1 << 20, 8, 16, None, None) == -1 and b"scratch" in lib.rm_last_error()
    assert lib.rm_tile_score_from_ray_cost(16, 0, 1, 1, 1, 8, 16, 16, None) == -1
    assert lib.rm_tile_score_from_ray_cost(16, 10, 3, 3, 1, 8, 16, 16, None) == -1       # 10 tiles are no 3 x 3 grid
    cam64 = _abi.RmCamera(ray_positions=16, ray_directions=16, num_cameras=1, height=4, width=4, dtype=2)
    assert render(0, cam_=cam64) == -1 and b"camera dtype" in lib.rm_last_error()
    assert lib.rm_wave_tiles(1, 1080, 1920, _abi.FLAG_TILE8X8) == 135 * 240
    assert lib.rm_wave_tiles(2, 9, 65, 0) == (2 * 9 * 65 + 63) // 64
    assert lib.rm_render_traj_floats(1, 9, 65, 5, _abi.FLAG_TILE8X8) == 5 * 3 * 64 * (2 * 9) and lib.rm_render_traj_floats(1, 9, 65, 0, 0) == 0


def _validate(rows, n_params, n_derived, stack, slots):
    from ray_marching_amd import _abi
    prog = np.asarray(rows, dtype=np.int32).reshape(-1, 4)
    return _abi.lib.rm_validate_program(prog.ctypes.data, prog.shape[0], n_params, n_derived, stack, slots)


def test_program_validation():
    A = __import__("ray_marching_amd._abi", fromlist=["x"])
    assert _validate([[A.OP_SPHERE, 0, 0, 0]], 1, 0, 0, 0) == 0
    assert _validate([[A.OP_SPHERE, 1, 0, 0]], 1, 0, 0, 0) == -2            # parameter out of range
    assert _validate([[99, 0, 0, 0]], 1, 0, 0, 0) == -2                     # bad opcode
    assert _validate([[A.OP_UNION_BEGIN, 0, 0, 0], [A.OP_SPHERE, 0, 0, 0]], 1, 0, 2, 1) == -2   # unbalanced
    assert _validate([[A.OP_SPHERE, 0, 0, 0], [A.OP_SPHERE, 0, 0, 0]], 1, 0, 0, 0) == -2        # two values
    good = [[A.OP_UNION_BEGIN, 0, 0, 0], [A.OP_SPHERE, 0, 0, 0], [A.OP_FOLD_MIN, 0, 0, 0],
            [A.OP_PLANE, 0, 0, 0], [A.OP_FOLD_MIN, 0, 1, 0], [A.OP_UNION_END, 0, 0, 2]]
    assert _validate(good, 1, 0, 2, 2) == 0
    assert _validate(good, 1, 0, 1, 2) == -2                                # stack too small
    assert _validate(good, 1, 0, 2, 1) == -2                                # slot out of range
    assert _validate([[A.OP_LINE, 0, 7, 0]], 7, 6, 0, 0) == 0
    assert _validate([[A.OP_LINE, 0, 5, 0]], 7, 6, 0, 0) == -2              # derived block overlaps raw params


def test_compiler_lowers_reference_scenes():
    from ray_marching_amd.compiler import compile_scene
    from ray_marching_amd.scene import scene_registry as R
    cs2 = compile_scene(R.make_test_scene2())
    # parameter names of make_test_scene2
    assert cs2.leaf_names == ["sdfs.0.radius", "sdfs.0.sdf.halfsides", "sdfs.1.sdfs.0.radius",
                              "sdfs.1.sdfs.1.radius1", "sdfs.1.sdfs.1.radius2", "sdfs.1.sdfs.2.start",
                              "sdfs.1.sdfs.2.end", "sdfs.1.sdfs.2.radius"]
    assert (cs2.n_params, cs2.n_derived, cs2.n_slots) == (14, 11, 6)       # 6 capsule constants + 1 cull bound {c, K, slope}
    assert cs2.leaf_names == [n for n, _ in O.spec_parameters(O.scene_test2())]
    cs1 = compile_scene(R.make_test_scene())
    assert cs1.leaf_names == [n for n, _ in O.spec_parameters(O.scene_test1())]
    assert cs1.leaf_names[0] == "blend_k" and cs1.n_params == 36
    csc = compile_scene(R.make_closed_test_scene())
    assert csc.leaf_names == [n for n, _ in O.spec_parameters(O.scene_test1_closed())] and csc.n_params == 40
    cs5 = compile_scene(R.make_many_primitive_scene(32))
    assert cs5.leaf_names == [n for n, _ in O.spec_parameters(O.scene_many(32))]
    # parameter VALUES of the factories equal the oracle's (same constants / same seed)
    for factory, spec in ((R.make_test_scene2, O.scene_test2()), (R.make_closed_test_scene, O.scene_test1_closed()),
                          (lambda: R.make_many_primitive_scene(32), O.scene_many(32))):
        for (n1, p1), (n2, p2) in zip(factory().named_parameters(), O.spec_parameters(spec)):
            assert n1 == n2 and torch.equal(p1.detach(), p2), n1
    # packing is differentiable plumbing and follows the offsets
    scene = R.make_test_scene2()
    cs = compile_scene(scene)
    flat = cs.pack_params("cpu")
    assert flat.requires_grad and flat.shape == (14,)
    assert torch.equal(flat[cs.leaf_offsets[5]:cs.leaf_offsets[5] + 3].detach(), scene.sdfs[1].sdfs[2].start.detach())
    # ... and always reflects the live values: in-place edits, optimiser steps, and writes through .data (which
    # do not bump the version counter) -- there is no cache to go stale
    with torch.no_grad():
        scene.sdfs[0].radius.add_(0.5)
        assert cs.pack_params("cpu")[0].item() == pytest.approx(0.6)
        scene.sdfs[0].radius.data.fill_(2.0)
        assert cs.pack_params("cpu")[0].item() == 2.0
    opt = torch.optim.SGD(scene.parameters(), lr=1.0)
    scene.sdfs[0].radius.grad = torch.tensor(0.5)
    opt.step()
    assert cs.pack_params("cpu")[0].item() == 1.5
    # the in-kernel gather table follows the parameter storages: same object while they stay put, rebuilt when a
    # parameter is re-allocated; rows = {pointer, element | dtype << 32} per float
    t1 = cs.param_table("cpu")
    assert t1 is cs.param_table("cpu") and t1.shape == (2 * cs.n_params,)
    rows = t1.view(-1, 2)
    assert rows[0, 0].item() == scene.sdfs[0].radius.data_ptr() and rows[0, 1].item() == 0
    off = cs.leaf_offsets[5]
    assert rows[off + 2, 0].item() == scene.sdfs[1].sdfs[2].start.data_ptr() and rows[off + 2, 1].item() == 2
    scene.sdfs[0].radius.data = torch.tensor(0.25)
    t2 = cs.param_table("cpu")
    assert t2 is not t1 and t2.view(-1, 2)[0, 0].item() == scene.sdfs[0].radius.data_ptr()
    scene.half()
    assert (cs.param_table("cpu").view(-1, 2)[:, 1] >> 32 == 1).all()           # RM_DTYPE_F16
    scene.double()
    assert cs.param_table("cpu") is None                                          # not readable in place: caller packs


def test_modules_deepcopy_and_pickle_after_compile(tmp_path):
    """the compile cache (ctypes library handles, device tensors) must not sit in the module's state:
    copy.deepcopy (EMA copies) and torch.save / torch.load of a whole scene or RenderLoop work after a compile."""
    import copy
    import pickle
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.scene import scene_registry as R
    scene = R.make_closed_test_scene()
    cs = compiled_for(scene)
    cs.lib()                                   # resolve the library handle like a first render does
    loop = RenderLoop(scene, num_cameras=1, px_width=8, px_height=6)
    loop.normals.tetra()
    loop._f32_cache["x"] = ("k", torch.zeros(1))
    twin = copy.deepcopy(scene)
    assert [n for n, _ in twin.named_parameters()] == [n for n, _ in scene.named_parameters()]
    assert compiled_for(twin) is not cs and compiled_for(twin).signature == cs.signature
    loop2 = copy.deepcopy(loop)
    assert loop2._f32_cache == {} and loop2.scene is not scene
    path = tmp_path / "loop.pt"
    torch.save(loop, path)
    loop3 = torch.load(path, weights_only=False)          # a file this test wrote itself
    assert torch.equal(loop3.camera.ray_directions, loop.camera.ray_directions)
    assert torch.equal(loop3.scene.blend_k if hasattr(loop3.scene, "blend_k") else loop3.scene.sdfs[0].blend_k,
                       scene.sdfs[0].blend_k)
    cs2 = pickle.loads(pickle.dumps(cs))                  # a CompiledScene travels as its program only
    assert cs2._lib is None and cs2._table == {} and (cs2.program == cs.program).all()


def test_compiler_cull_placement_and_evaluation_order(monkeypatch):
    """CULL_MIN: only in front of expensive, boundable children of a min-union; children that
    cannot be culled are EVALUATED first while tape slots keep the reference's child order (the reverse
    pass picks the first minimal slot = torch's argmin); every variant passes rm_validate_program."""
    import numpy as np
    from ray_marching_amd import _abi
    from ray_marching_amd.compiler import compile_scene
    from ray_marching_amd.scene import scene_registry as R
    from ray_marching_amd.scene.primitives import SDFPlane, SDFSphere
    from ray_marching_amd.scene.transformations import SDFUnion

    def rows(cs):
        return np.asarray(cs.program).reshape(-1, 4)

    def ops(cs):
        return rows(cs)[:, 0].tolist()

    # closed scene1 = Union([smooth group (expensive, bounded), room]): the room is evaluated first,
    # the group behind a test; slots: group = 0 (child 0), room = 1 (child 1)
    r = rows(compile_scene(R.make_closed_test_scene()))
    folds = [(int(x[2]), int(x[3])) for x in r if x[0] == _abi.OP_FOLD_MIN]
    culls = [i for i, x in enumerate(r) if x[0] == _abi.OP_CULL_MIN]
    assert folds[0] == (1, 0) and folds[1][0] == 0 and folds[1][1] > 0 and len(culls) == 1
    skip, slot = int(r[culls[0], 3]) >> 8, int(r[culls[0], 3]) & 255
    assert slot == 0 and r[culls[0] + skip, 0] == _abi.OP_FOLD_MIN and int(r[culls[0] + skip, 3]) == skip
    # a smooth union alone, cheap leaves, and unbounded children get no test
    assert _abi.OP_CULL_MIN not in ops(compile_scene(R.make_test_scene()))
    assert _abi.OP_CULL_MIN not in ops(compile_scene(SDFUnion([SDFSphere(0.3), SDFSphere(0.4)])))
    assert _abi.OP_CULL_MIN not in ops(compile_scene(SDFUnion([R.make_room(), SDFPlane()])))
    # knobs: RM_CULL=0 removes them, RM_CULL_MIN_COST=0 tests every boundable child but never the first one
    # evaluated and never an unbounded one, RM_CULL_REORDER=0 keeps the child order
    monkeypatch.setenv("RM_CULL", "0")
    assert _abi.OP_CULL_MIN not in ops(compile_scene(R.make_test_scene2()))
    monkeypatch.setenv("RM_CULL", "1")
    monkeypatch.setenv("RM_CULL_MIN_COST", "0")
    cs = compile_scene(SDFUnion([SDFSphere(0.3), SDFPlane(), SDFSphere(0.4), SDFSphere(0.5)]))
    r = rows(cs)
    assert [int(x[2]) for x in r if x[0] == _abi.OP_FOLD_MIN] == [1, 0, 2, 3]       # plane first, then child order
    assert sum(x[0] == _abi.OP_CULL_MIN for x in r) == 3 and cs.n_derived == 15
    monkeypatch.setenv("RM_CULL_REORDER", "0")
    r = rows(compile_scene(SDFUnion([SDFSphere(0.3), SDFPlane(), SDFSphere(0.4)])))
    assert [int(x[2]) for x in r if x[0] == _abi.OP_FOLD_MIN] == [0, 1, 2]
    assert sum(x[0] == _abi.OP_CULL_MIN for x in r) == 1                            # only the last sphere


def test_one_hip_runtime_whatever_the_import_order():
    """torch bundles its own libamdhip64.so; if librm_hip.so were opened first the process would map
    /opt/rocm's copy as well and our launches would fail with "no ROCm-capable device" (seen on the GPU box:
    build() then smoke() in one process).  _abi imports torch first: one runtime, in either order."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for first in ("import ray_marching_amd, torch", "import torch, ray_marching_amd",
                  "import __graft_entry__ as g; g.build_library(); import ray_marching_amd, torch"):
        code = (f"import sys; sys.path.insert(0, {root!r}); {first}\n"
                "paths = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}\n"
                "print(len(paths), sorted(paths))")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.split()[0] == "1", (first, r.stdout)


def test_foreign_module_is_rejected_loudly():
    from ray_marching_amd.compiler import compile_scene
    from ray_marching_amd.scene.transformations import SDFUnion
    with pytest.raises(TypeError, match="not a ray_marching_amd SDF node"):
        compile_scene(SDFUnion([torch.nn.Linear(3, 1)]))


def test_no_cpu_fallback():
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.rendering.ray_marching import SDFMarcher, SDFNormals
    from ray_marching_amd.scene.scene_registry import make_test_scene2
    scene = make_test_scene2()
    x = torch.zeros(4, 3)
    for call in (lambda: scene(x), lambda: SDFMarcher(scene)(x, x, 4), lambda: SDFNormals(scene)(x)):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            call()
    loop = RenderLoop(scene, px_width=8, px_height=8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        loop(torch.tensor([[1.0, 0, 0, 0]]), torch.zeros(1, 3))


def test_module_surface_mirrors_reference():
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.rendering.ray_marching import PinholeCamera, tetrahedron_constants
    from ray_marching_amd.rendering import shader as S
    from ray_marching_amd.scene.scene_registry import make_test_scene2
    h, w = 12, 16
    cam = PinholeCamera(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    origins, directions = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    assert torch.equal(cam.ray_positions, origins) and torch.equal(cam.ray_directions, directions)
    g = H.gold("f2_camera.npz")
    assert np.array_equal(cam.ray_positions.numpy(), g["ray_positions"])      # reference's own buffers
    taps, rel, inv = tetrahedron_constants(5e-2)
    o_taps, o_inv = O.tetra_constants(5e-2)
    assert torch.equal(taps, o_taps) and torch.equal(inv, o_inv)
    assert inv[0, 0].item() == pytest.approx(-12.2474, abs=1e-3)             # SURVEY A6 anchor
    loop = RenderLoop(make_test_scene2(), px_width=w, px_height=h)
    keys = set(loop.state_dict().keys())
    for k in ("camera.focus", "camera.theta", "camera.ray_positions", "camera.ray_directions", "camera.pixel_frames",
              "camera.quaternion_to_so3.pairs", "normals.offsets", "normals.relative_offsets",
              "normals.offsets_inverse", "shader.cyclic_cmap", "scene.sdfs.1.sdfs.2.start"):
        assert k in keys, k
    assert S.MODES == ["lambertian", "distance", "proximity", "vignette", "normal", "laplacian", "tangent", "spin"]
    cm = S.default_cyclic_cmap()
    assert cm.shape == (4096, 3) and cm.dtype == torch.float64 and 0.23 < cm.min() < 0.25 and cm.max() <= 1.0


def test_quaternion_helpers_match_oracle():
    from ray_marching_amd import quaternion as Q
    gen = torch.Generator().manual_seed(5)
    u, v = torch.randn(64, 3, generator=gen), torch.randn(64, 3, generator=gen)
    p, q = torch.randn(64, 4, generator=gen), torch.randn(64, 4, generator=gen)
    assert torch.equal(Q.cross_product(u, v), O.cross(u, v))
    assert torch.equal(Q.rotation(u, q), O.quat_rotate(u, q))
    assert torch.equal(Q.conjugate(q), O.quat_conj(q))
    assert torch.equal(Q.multiply(p, q), O.quat_multiply(p, q))      # the reference's op order (quaternion.py:38-46), bit for bit
    torch.testing.assert_close(Q.QuaternionToSO3()(q), O.quat_to_so3(q), rtol=1e-6, atol=1e-6)
    w = Q.to_versor(torch.tensor([[0.1, 0.2, 0.2]]))
    assert w.shape == (1, 4) and w.norm().item() == pytest.approx(1.0, abs=1e-6)


def test_specialisation_codegen_is_deterministic():
    from ray_marching_amd import specialize
    from ray_marching_amd.compiler import compile_scene
    from ray_marching_amd.scene.scene_registry import make_test_scene2
    a, b = compile_scene(make_test_scene2()), compile_scene(make_test_scene2())
    assert specialize.scene_hash(a) == specialize.scene_hash(b)
    hdr = specialize.code_header(a)
    assert "static constexpr int n = 15" in hdr and hdr.count("{") == 17      # 14 node instructions + 1 CULL_MIN
    assert specialize.static_backward(a)
    from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
    assert not specialize.static_backward(compile_scene(make_many_primitive_scene(32)))


def test_headless_stand_ins_follow_the_reference_contracts(tmp_path):
    """PosePlayer integrates poses like EventAggregator.get_state (control.py:150-165); FrameSink.draw
    enforces Window.draw's input contract ([H,W,4] fp32 contiguous; window.py:146-174)."""
    from ray_marching_amd.headless import FrameSink, PosePlayer, to_rgba
    p = PosePlayer([(0.0, 0.0, 1.0)], [(1.0, 0.0, 0.0, 0.0)], marching_steps=32, velocity=(1.0, 0.0, 0.0),
                   angular_velocity=(0.0, 0.0, 0.1), mode_every=2)
    pos, q, mode, degree, steps, save = p.get_state()
    assert pos.shape == (1, 3) and q.shape == (1, 4) and (mode, degree, steps, save) == (0, 2, 32, False)
    # oracle of the update: position += rot(v*0.1, q_old); q = normalize(q_old (x) versor(w*0.25))
    q_old = torch.tensor([[1.0, 0.0, 0.0, 0.0]])
    want_pos = O.quat_rotate(torch.tensor([[0.1, 0.0, 0.0]]), q_old) + torch.tensor([[0.0, 0.0, 1.0]])
    w = torch.tensor([[0.0, 0.0, 0.025]])
    versor = torch.cat([(1 - w.pow(2).sum(-1, keepdim=True)).sqrt(), w], -1)
    want_q = torch.nn.functional.normalize(O.quat_multiply(q_old, versor), dim=-1)
    torch.testing.assert_close(pos, want_pos); torch.testing.assert_close(q, want_q)
    assert p.get_state()[2] == 1                      # mode cycles every 2 frames
    img = torch.rand(2, 6, 8, 3)
    rgba = to_rgba(img)
    assert rgba.shape == (6, 8, 4) and rgba.is_contiguous() and bool((rgba[..., 3] == 1).all())
    sink = FrameSink(8, 6, out_dir=str(tmp_path))
    sink.draw(rgba)
    assert torch.equal(sink.latest(), rgba)
    data = open(tmp_path / "frame_00000.ppm", "rb").read()
    assert data.startswith(b"P6 8 6 255\n") and len(data) == 11 + 8 * 6 * 3
    with pytest.raises(ValueError):
        sink.draw(rgba[:, :, :3])
    with pytest.raises(ValueError):
        sink.draw(rgba.double())
    # PNG (decoded again with zlib: signature, IHDR, one IDAT of filter-0 rows) and the raw float32 frame
    import struct
    import zlib
    png_sink = FrameSink(8, 6, out_dir=str(tmp_path / "png"), file_format="png")
    png_sink.draw(rgba)
    data = open(tmp_path / "png" / "frame_00000.png", "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    chunks, at = [], 8
    while at < len(data):
        n, tag = struct.unpack(">I", data[at:at + 4])[0], data[at + 4:at + 8]
        body = data[at + 8:at + 8 + n]
        assert struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        chunks.append((tag, body)); at += 12 + n
    assert [t for t, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
    assert struct.unpack(">IIBBBBB", chunks[0][1]) == (8, 6, 8, 2, 0, 0, 0)
    rows = zlib.decompress(chunks[1][1])
    want = (rgba[..., :3].clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
    for y in range(6):
        assert rows[y * 25] == 0 and rows[y * 25 + 1:(y + 1) * 25] == want[y].numpy().tobytes()
    raw_sink = FrameSink(8, 6, out_dir=str(tmp_path / "raw"), file_format="raw")
    raw_sink.draw(rgba)
    assert open(tmp_path / "raw" / "frame_00000.raw", "rb").read() == rgba.numpy().tobytes()
    with pytest.raises(ValueError):
        FrameSink(8, 6, file_format="jpeg")


def test_background_specialisation_policy(monkeypatch, tmp_path):
    """"auto" policy: a scene that keeps running on the interpreter gets ONE background hipcc build after
    RM_SPECIALIZE_AFTER launches and is switched over when the library exists (no GPU involved)."""
    from ray_marching_amd import _abi, specialize
    from ray_marching_amd.compiler import compile_scene
    from ray_marching_amd.scene.primitives import SDFDisk
    from ray_marching_amd.scene.transformations import SDFRounding
    monkeypatch.setattr(specialize, "SPEC_DIR", str(tmp_path))
    monkeypatch.setenv("RM_SPECIALIZE", "auto")
    monkeypatch.setenv("RM_SPECIALIZE_AFTER", "3")
    specialize._loaded.clear(); specialize._uses.clear()
    cs = compile_scene(SDFRounding(SDFDisk(0.7), 0.05))        # a topology nobody prebuilt
    assert cs.lib() is _abi.lib and cs.lib() is _abi.lib        # launches 1, 2: interpreter, no build
    assert specialize._builder["thread"] is None
    assert cs.lib() is _abi.lib                                 # launch 3 starts the background build
    assert specialize._builder["thread"] is not None
    specialize.wait_for_background_build(180)
    assert os.path.isfile(specialize.lib_path(cs))
    assert cs.lib() is not _abi.lib and cs.specialised          # picked up on the next launch
    monkeypatch.setenv("RM_SPECIALIZE", "off")
    assert compile_scene(SDFRounding(SDFDisk(0.7), 0.05)).lib() is _abi.lib
    specialize._loaded.clear(); specialize._uses.clear()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
    leg may touch it.  The product package must not (no CPU fallback can hide behind it), and it must not
    read the reference tree either."""
    import ast
    pkg = os.path.join(ROOT, "ray_marching_amd")
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if not name.endswith(".py"):
                continue
            path = os.path.join(dirpath, name)
            src = open(path).read()
            tree = ast.parse(src)
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                elif isinstance(node, ast.ImportFrom):
                    mods = [node.module or ""]
                for m in mods:
                    assert not m.split(".")[0] == "oracle", f"{path} imports {m}"
            assert "/root/reference" not in src, f"{path} mentions the reference tree"
    # bench.py: the oracle appears only inside cpu_baseline(); __graft_entry__: only inside smoke()
    for fname, allowed in (("bench.py", "cpu_baseline"), ("__graft_entry__.py", "smoke")):
        tree = ast.parse(open(os.path.join(ROOT, fname)).read())
        for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            uses = any(isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle" for n in ast.walk(fn))
            assert not uses or fn.name == allowed, f"{fname}:{fn.name} imports the oracle"
        top_level = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
        assert not any((getattr(n, "module", "") or "").startswith("oracle") for n in top_level)


def test_auto_kernel_choice_state_machine():
    """RenderLoop(regen="auto"): probe in the first frame of a cycle, the kernel in use timed in the second; the pools
    take over when > 3 % faster and are then checked against the tile kernel every cycle, which takes over again as
    soon as it is faster; with the tile kernel in use the pools are looked at every fourth cycle.  Driven here with
    stand-in timing events."""
    from ray_marching_amd.control import RenderLoop

    class Ev:
        def __init__(self, ms, ready=True):
            self.ms, self.ready = ms, ready

        def query(self):
            return self.ready

        def elapsed_time(self, other):
            return other.ms

    loop = RenderLoop.__new__(RenderLoop)                     # the state machine needs no module state
    loop._choice_state, loop.adaptive_order = {}, 4
    speed = {"tile": 400.0, "regen": 300.0}

    def frame():
        regen, record, sink = loop._choose_kernel_for("k")
        if sink is not None:
            sink.append((Ev(0.0), Ev(speed["regen" if regen else "tile"])))
        return regen, record, sink is not None

    log = [frame() for _ in range(4 * 12)]
    assert log[0] == (True, True, True) and log[1] == (False, False, True)      # probe of the pools, then the tile kernel timed
    assert log[2][0] is True and log[3] == (True, False, False)                 # the pools were faster: in use from frame 2 on
    assert log[4] == (False, True, True) and log[5] == (True, False, True)      # next cycle: the tile kernel is looked at
    # while the pools are in use the tile kernel is looked at again and again: every second cycle where it is > 15 % behind
    probes = [i for i, (regen, record, timed) in enumerate(log) if record]
    assert probes == [0, 4] + list(range(12, 48, 8)), probes
    assert all(regen for i, (regen, _, _) in enumerate(log) if i >= 2 and i not in probes)
    # the scene changes: the tile kernel becomes the faster one; the next probe finds out and it takes over at once ...
    speed.update(tile=200.0)
    log2 = [frame() for _ in range(4 * 20)]
    probes2 = [i for i, (regen, record, timed) in enumerate(log2) if record]
    assert probes2[0] <= 4 and log2[probes2[0] + 2][0] is False and log2[-1][0] is False
    # ... and from then on the pools are only looked at every fourth cycle
    assert probes2[1] - probes2[0] == 4 and all(b - a == 16 for a, b in zip(probes2[1:], probes2[2:])), probes2
    # a measurement that is not ready yet decides nothing (and blocks nothing)
    loop._choice_state.clear()
    r, rec, sink = loop._choose_kernel_for("k")
    pending = Ev(1.0, ready=False)
    sink.append((Ev(0.0), pending))
    r1, _, sink1 = loop._choose_kernel_for("k")
    sink1.append((Ev(0.0), Ev(500.0)))
    assert loop._choose_kernel_for("k")[0] is False and loop._choice_state["k"]["regen"] is False
    pending.ready, pending.ms = True, 100.0
    assert loop._choose_kernel_for("k")[0] is True


def test_auto_kernel_choice_survives_a_host_far_ahead_of_the_gpu():
    """Timing events that complete dozens of frames after they were recorded (the host enqueues faster than the GPU
    renders) must still lead to a decision: measurements in flight are waited for, not started over."""
    from ray_marching_amd.control import RenderLoop

    class Ev:
        def __init__(self, ms):
            self.ms, self.ready = ms, False

        def query(self):
            return self.ready

        def elapsed_time(self, other):
            return other.ms

    loop = RenderLoop.__new__(RenderLoop)
    loop._choice_state, loop.adaptive_order = {}, 4
    in_flight = []
    for frame in range(200):
        regen, record, sink = loop._choose_kernel_for("k")
        if sink is not None:
            ev = Ev(300.0 if regen else 400.0)
            sink.append((Ev(0.0), ev))
            in_flight.append((frame, ev))
        for f0, ev in in_flight:                  # the GPU is 40 frames behind
            if frame - f0 >= 40:
                ev.ready = True
    assert loop._choice_state["k"]["regen"] is True and len(loop._choice_state["k"]["log"]) >= 2


def test_torch_compile_steps_over_the_ctypes_launch():
    """main.py:44 wraps the RenderLoop in torch.compile(mode='max-autotune').  The forwards are marked
    torch.compiler.disable, so Dynamo does not trace into the ctypes call: the eager code runs -- seen here, without a
    GPU, by the eager path's own refusal of CPU tensors arriving through the compiled wrapper."""
    from ray_marching_amd.control import RenderLoop
    from ray_marching_amd.scene.scene_registry import make_test_scene2
    loop = torch.compile(RenderLoop(make_test_scene2(), px_width=16, px_height=8), mode="max-autotune")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        loop(torch.tensor([[1.0, 0.0, 0.0, 0.0]]), torch.zeros(1, 3), 0, 1, 32)
    scene = torch.compile(make_test_scene2())
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        scene(torch.zeros(4, 3))


def test_auto_kernel_choice_needs_two_probes_for_a_narrow_loss():
    """Pools in use: a probe that shows the tile kernel ahead by less than 10 % only raises a doubt; the second one in a
    row hands over; a clear margin hands over at once; a probe in between that favours the pools clears the doubt."""
    from ray_marching_amd.control import RenderLoop

    class Ev:
        def __init__(self, ms):
            self.ms = ms

        def query(self):
            return True

        def elapsed_time(self, other):
            return other.ms

    loop = RenderLoop.__new__(RenderLoop)
    loop._choice_state, loop.adaptive_order = {}, 4
    speed = {"tile": 400.0, "regen": 300.0}

    def run(frames):
        out = []
        for _ in range(frames):
            regen, record, sink = loop._choose_kernel_for("k")
            if sink is not None:
                sink.append((Ev(0.0), Ev(speed["regen" if regen else "tile"])))
            out.append(regen)
        return out

    assert run(12)[-1] is True                         # the pools took over
    speed.update(tile=290.0)                           # the tile kernel 3 % ahead: first probe = doubt only
    seq = run(4 * 6)
    first_tile = seq.index(False, 4) if False in seq[4:] else None
    st = loop._choice_state["k"]
    assert st["regen"] is False and first_tile is not None
    narrow = [e for e in st["log"] if e[1] and e[2] == 290.0]
    assert len(narrow) == 2, st["log"]                 # two probes of the tile kernel were needed
    loop._choice_state.clear()
    speed.update(tile=400.0)
    run(12)
    speed.update(tile=250.0)                           # 17 % ahead: at once
    run(12)
    st = loop._choice_state["k"]
    assert st["regen"] is False and len([e for e in st["log"] if e[1] and e[2] == 250.0]) == 1, st["log"]


def test_auto_kernel_choice_looks_again_after_a_cold_first_probe():
    """The very first probe of the pools runs them without a dealing order and with first-use allocations; when it loses,
    the pools are looked at again in the NEXT cycle (then every fourth), so a scene where they are the faster kernel
    is not rendered 64 frames on the slower one."""
    from ray_marching_amd.control import RenderLoop

    class Ev:
        def __init__(self, ms):
            self.ms = ms

        def query(self):
            return True

        def elapsed_time(self, other):
            return other.ms

    loop = RenderLoop.__new__(RenderLoop)
    loop._choice_state, loop.adaptive_order = {}, 4
    pools_ms = iter([505.0] + [320.0] * 100)          # cold, then what they really take
    probes, in_use = [], []
    for frame in range(4 * 8):
        regen, record, sink = loop._choose_kernel_for("k")
        if sink is not None:
            sink.append((Ev(0.0), Ev(next(pools_ms) if regen else 420.0)))
        if record:
            probes.append(frame)
        in_use.append(regen)
    assert probes[:2] == [0, 4], probes               # second look one cycle later, not four
    assert all(r for f, r in enumerate(in_use) if f >= 7 and f not in probes), in_use     # ... after which the pools are in use (probe frames run the tile kernel)


def test_abi_header_is_plain_c():
    """include/rm_abi.h is the boundary a maintainer binds from C, cgo, JNI or ctypes: it must compile as C99 and as
    C++11 on its own, without HIP or torch headers, and a C program must be able to name every entry point."""
    import shutil
    import subprocess
    import tempfile
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    header = os.path.join(ROOT, "include", "rm_abi.h")
    for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++11", "c++")):
        r = subprocess.run([cc, std, "-fsyntax-only", "-Wall", "-Werror", "-x", lang, header], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "use.c")
        with open(src, "w") as f:
            f.write('#include "rm_abi.h"\nint main(void) {\n  void* p[] = {' +
                    ", ".join(f"(void*){name}" for name in declared_functions()) +
                    "};\n  RmScene s; RmCamera c; RmTetra t; RmParamRef r;\n  (void)s; (void)c; (void)t; (void)r;\n"
                    "  return (int)(sizeof(p) / sizeof(p[0])) + RM_ABI_VERSION + RM_OP_CULL_LSE + RM_DTYPE_RGBA_F32 + RM_WORK_WORDS > 0 ? 0 : 1;\n}\n")
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"), "-c", src, "-o", os.path.join(tmp, "use.o")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def test_ctypes_structs_have_the_layout_of_the_header():
    """The ctypes mirrors in _abi.py (what every launch passes by pointer) against the C compiler's own layout of
    include/rm_abi.h: size and every field offset of RmScene, RmCamera, RmTetra and RmParamRef (ABI v12 / v13 added
    RmScene.block / block_out / block_cache at the end)."""
    import shutil
    import subprocess
    import tempfile
    from ray_marching_amd import _abi
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    structs = {"RmScene": _abi.RmScene, "RmCamera": _abi.RmCamera, "RmTetra": _abi.RmTetra}
    lines = []
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({name}, {field}));')
        lines.append('  printf("\\n");')
    lines.append('  printf("RmParamRef %zu %zu %zu %zu\\n", sizeof(RmParamRef), offsetof(RmParamRef, base), offsetof(RmParamRef, elem), offsetof(RmParamRef, dtype));')
    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "layout.c"), os.path.join(tmp, "layout")
        with open(src, "w") as f:
            f.write('#include <stddef.h>\n#include <stdio.h>\n#include "rm_abi.h"\nint main(void) {\n' + "\n".join(lines) + "\n  return 0;\n}\n")
        r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        out = subprocess.run([exe], capture_output=True, text=True).stdout.splitlines()
    got = {line.split()[0]: [int(x) for x in line.split()[1:]] for line in out}
    for name, cls in structs.items():
        want = [C.sizeof(cls)] + [getattr(cls, field).offset for field, _ in cls._fields_]
        assert got[name] == want, (name, got[name], want)
    assert [f for f, _ in _abi.RmScene._fields_[-3:]] == ["block", "block_out", "block_cache"]
    # the pointer table rows compiler.py writes: {int64 pointer, int64 elem | dtype << 32} = RmParamRef on a little-endian host
    assert got["RmParamRef"] == [16, 0, 8, 12]
