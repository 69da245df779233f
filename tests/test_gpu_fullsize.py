"""BASELINE configs 3 and 5 at their own size and iteration count, and the native fp16 I/O path.

configs[2]: make_test_scene2, 3840x2160, 256 march steps, module .to(float16)  (main.py:20-26 runs float16)
configs[4]: 32-primitive smooth-union scene in a room, 7680x4320, 256 steps, 8 row bands of 540 rows

The oracle cannot render these frames in the time a test has, so the full frames are checked through
size-independent properties (early-out on/off, row-band reassembly, interpreter == specialised kernels, finite
values) and a strided pixel sample is compared with the oracle at the full step count, bit for bit.
"""
import os

import pytest
import torch

from oracle import sdf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _fp16_oracle_inputs(spec, n, w, h):
    """What a reference module cast with .to(float16) holds, as fp32 tensors: fp16-rounded parameters, camera
    buffers and tetrahedron constants (the kernels read fp16 storage and compute in fp32)."""
    r = lambda x: x.half().float()
    bufs = tuple(r(b) for b in O.camera_buffers(n, w, h, H.PX * h, H.PX * w, H.PX * h))
    tetra = tuple(r(c) for c in O.tetra_constants(H.EPS))
    return O.map_spec(spec, r), bufs, tetra


@pytest.mark.parametrize("shape", [(1, 90, 160), (2, 33, 47)])
def test_fp16_io_equals_fp32_arithmetic_on_fp16_storage(shape):
    """RenderLoop.to(float16): fp16 camera buffers, parameters, pose and image, fp32 arithmetic in between.
    That is exactly the fp32 oracle evaluated on the fp16-rounded inputs with its result rounded once to fp16 --
    compared bit for bit, every shader mode (the reference's own fp16 path rounds after EVERY ATen op and sits
    up to 0.66 away from its fp32 path, see test_fp16_io_config3_numerics for that statistical comparison)."""
    n, h, w = shape
    steps = 48
    spec = O.scene_test2()
    loop = H.make_loop(H.spec_to_module(spec), h, w, n=n).to(torch.float16)
    assert loop.camera.ray_positions.dtype == torch.float16 and loop.scene.sdfs[0].radius.dtype == torch.float16
    cmap16 = loop.shader.cyclic_cmap                     # .to(float16) cast the registered buffer like the reference's
    assert cmap16.dtype == torch.float16
    spec16, bufs, tetra = _fp16_oracle_inputs(spec, n, w, h)
    gen = torch.Generator().manual_seed(5)
    q = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0, 0.0, 0.0]]) + 0.1 * torch.randn(n, 4, generator=gen), dim=-1).half()
    t = (torch.tensor([[0.0, 0.0, -3.0]]) + 0.2 * torch.randn(n, 3, generator=gen)).half()
    for mode in range(8):
        if mode == 3 and n > 1:
            continue        # the reference's vignette shader only broadcasts for one camera (shader.py:64)
        with torch.no_grad(), O.math_mode("restated"):
            want = O.render(spec16, bufs, q.float(), t.float(), mode, 2, steps, H.EPS, cmap=cmap16.float().cpu(),
                            tetra=tetra).half()
            got = loop(q.to(DEV), t.to(DEV), mode, 2, steps)
        assert got.dtype == torch.float16 and got.shape == (n, h, w, 3)
        assert H.report(f"fp16 mode {mode}", got, want)[0] == 0.0, mode


def test_fp16_frame_is_one_kernel_and_no_cast_passes():
    """The fp16 frame issues the workspace init and ONE frame kernel -- no torch cast / copy kernels (the
    round-1 plumbing converted buffers, pose and image in separate passes)."""
    from torch.profiler import ProfilerActivity, profile
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), 120, 160).to(torch.float16)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV).half()
    t = torch.tensor([[0.0, 0.0, -3.0]], device=DEV).half()
    with torch.no_grad():
        for _ in range(3):
            loop(q, t, 4, 1, 32)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
            img = loop(q, t, 4, 1, 32)
            torch.cuda.synchronize()
    assert img.dtype == torch.float16
    kernels = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    if not kernels:
        pytest.skip("the profiler recorded no device activity on this box")
    print("kernels of one fp16 frame:", kernels)
    assert sum("k_render_fwd" in k for k in kernels) == 1
    others = [k for k in kernels if "k_render_fwd" not in k and "k_minmax_init" not in k and "Memset" not in k]
    assert not others, others


def test_config3_full_size_fp16(monkeypatch):
    """configs[2] at 3840x2160, 256 steps, float16 module."""
    from ray_marching_amd import specialize
    from ray_marching_amd.compiler import compiled_for
    h, w, steps = 2160, 3840, 256
    spec = O.scene_test2()
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV).half()
    t = torch.tensor([[0.0, 0.0, -3.0]], device=DEV).half()
    frames = {}
    for path in ("auto", "off"):
        monkeypatch.setenv("RM_SPECIALIZE", path)
        specialize._loaded.clear()
        loop = H.make_loop(H.spec_to_module(spec), h, w).to(torch.float16)
        assert compiled_for(loop.scene).specialised == (path == "auto")
        with torch.no_grad():
            frames[path] = {m: loop(q, t, m, 1, steps) for m in (4, 0)}
        if path == "auto":
            loop_full = H.make_loop(H.spec_to_module(spec), h, w, early_out=False).to(torch.float16)
            with torch.no_grad():
                for m in (4, 0):
                    a = frames[path][m]
                    assert a.dtype == torch.float16 and a.shape == (1, h, w, 3) and torch.isfinite(a).all()
                    assert torch.equal(a, loop_full(q, t, m, 1, steps)), "early-out changed pixels"
                    bands = [loop(q, t, m, 1, steps, rows=(r, r + h // 8)) for r in range(0, h, h // 8)]
                    assert torch.equal(torch.cat(bands, dim=1), a), "8 row bands do not reassemble to the frame"
    specialize._loaded.clear()
    for m in (4, 0):
        assert torch.equal(frames["auto"][m], frames["off"][m]), "specialised kernels != interpreter"
    # every 32nd pixel of the frame vs the oracle at the full 256 steps (fp32 arithmetic on the fp16 storage)
    stride = 32
    spec16, bufs, tetra = _fp16_oracle_inputs(spec, 1, w, h)
    sub = tuple(b[:, ::stride, ::stride].contiguous() for b in bufs)
    for m in (4, 0):
        with torch.no_grad():
            want = O.render(spec16, sub, q.float().cpu(), t.float().cpu(), m, 1, steps, H.EPS, tetra=tetra).half()
        assert H.report(f"config 3 mode {m}", frames["auto"][m][:, ::stride, ::stride], want)[0] == 0.0


def test_config5_full_size_band_and_frame(monkeypatch):
    """configs[4]: the 7680x4320 frame of the 32-primitive scene at 256 steps, as 8 bands of 540 rows."""
    from ray_marching_amd import specialize
    from ray_marching_amd.compiler import compiled_for
    from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
    h, w, steps, bands = 4320, 7680, 256, 8
    rows = h // bands
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -4.5]], device=DEV)
    band = (3 * rows, 4 * rows)                           # rows 1620 .. 2160: through the middle of the objects
    out = {}
    for path in ("auto", "off"):
        monkeypatch.setenv("RM_SPECIALIZE", path)
        specialize._loaded.clear()
        loop = H.make_loop(make_many_primitive_scene(32), h, w)
        assert compiled_for(loop.scene).specialised == (path == "auto")
        with torch.no_grad():
            out[path] = loop(q, t, 4, 1, steps, rows=band)
        if path == "auto":
            a = out[path]
            assert a.shape == (1, rows, w, 3) and torch.isfinite(a).all()
            loop_full = H.make_loop(make_many_primitive_scene(32), h, w, early_out=False)
            with torch.no_grad():
                assert torch.equal(a, loop_full(q, t, 4, 1, steps, rows=band)), "early-out changed pixels"
                # the whole 8K frame in one launch == the 8 bands a node renders, reassembled
                whole = loop(q, t, 4, 1, steps)
                assert torch.equal(whole[:, band[0]:band[1]], a)
                for b in (0, 7):
                    assert torch.equal(loop(q, t, 4, 1, steps, rows=(b * rows, (b + 1) * rows)), whole[:, b * rows:(b + 1) * rows])
                # a globally normalised shader over bands: min/max folded across launches through the hook
                del whole
    specialize._loaded.clear()
    assert torch.equal(out["auto"], out["off"]), "specialised kernels != interpreter"
    # every 32nd pixel of the band vs the oracle at 256 steps (host-independent exp/log), bit for bit
    stride = 32
    bufs = O.camera_buffers(1, w, h, H.PX * h, H.PX * w, H.PX * h)
    sub = tuple(b[:, band[0]:band[1]:stride, ::stride].contiguous() for b in bufs)
    with torch.no_grad(), O.math_mode("restated"):
        want = O.render(O.scene_many(32), sub, q.cpu(), t.cpu(), 4, 1, steps, H.EPS)
    assert H.report("config 5 band", out["auto"][:, ::stride, ::stride], want)[0] == 0.0


def test_main_py_loop_through_pose_player_and_frame_sink(tmp_path):
    """The reference's frame loop (main.py:53-88) on a display-less node: PosePlayer stands in for
    EventAggregator.get_state(), RenderLoop renders, FrameSink stands in for Window.draw and receives
    F.pad(images.mean(0).float(), [0,1], value=1.0) -- contiguous [H,W,4] fp32 (torchwindow/window.py:146-174)."""
    import torch.nn.functional as F
    from ray_marching_amd.headless import FrameSink, PosePlayer, to_rgba
    from ray_marching_amd.scene.scene_registry import make_test_scene2
    n, h, w = 2, 45, 80
    loop = H.make_loop(make_test_scene2(), h, w, n=n)
    # the reference's start pose (main.py:46-49) for camera 0, a second camera outside the torus; flying backwards
    # while yawing, shader mode advanced every frame (the scroll wheel): all 8 modes in 8 frames
    events = PosePlayer(initial_position=[[0.0, 0.0, 1.0], [0.3, -0.2, -3.0]],
                        initial_orientation=[[1.0, 0.0, 0.0, 0.0], [0.9, 0.1, -0.3, 0.2]], marching_steps=32, mode=0,
                        degree=2, velocity=(0.0, 0.0, -1.5), angular_velocity=(0.0, 0.2, 0.0), mode_every=1, device=DEV)
    window = FrameSink(width=w, height=h, out_dir=str(tmp_path))
    bufs = O.camera_buffers(n, w, h, H.PX * h, H.PX * w, H.PX * h)
    cmap = loop.shader.cyclic_cmap.cpu()
    seen = set()
    with torch.no_grad():
        for i in range(8):
            positions, orientations, mode, degree, marching_steps, _ = events.get_state()      # main.py:56-63
            if mode % 8 == 3:
                continue    # vignette: the reference itself raises with two cameras (shader.py:64 broadcast)
            images = loop(orientations, positions, mode, degree, marching_steps)                  # main.py:65-71
            window.draw(F.pad(images.mean(0).float(), [0, 1], value=1.0))                         # main.py:78-84
            seen.add(mode % 8)
            with O.math_mode("restated"):
                want = O.render(O.scene_test2(), bufs, orientations.cpu(), positions.cpu(), mode, degree,
                                marching_steps, H.EPS, cmap=cmap)
            want = F.pad(want.mean(0).float(), [0, 1], value=1.0)
            got = window.latest()
            assert got.shape == (h, w, 4) and got.dtype == torch.float32 and got.is_contiguous()
            assert torch.equal(got, to_rgba(images).cpu())
            assert H.report(f"sink frame {i} (mode {mode % 8})", got, want)[0] == 0.0
    window.close()
    assert window.frames == 7 and seen == set(range(8)) - {3}
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".ppm")) == [f"frame_{i:05d}.ppm" for i in range(7)]


def test_parking_build_changes_no_pixel(tmp_path):
    """The opt-in -DRM_PARKING build (rays that never settle are handed to a dense second kernel, k_render_parked)
    renders the reference's default pose -- a third of the wave tiles park some of their rays there -- bit for bit
    like the full-length march.  Runs in a child process: the variant libraries are built there (hipcc, ~30 s)."""
    import subprocess
    import sys
    from ray_marching_amd import specialize
    if specialize._hipcc() is None or not os.path.exists(specialize._hipcc()):
        pytest.skip("hipcc not available on this box")
    code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from tests import helpers as H
from oracle import sdf_oracle as O
from ray_marching_amd import _abi
h, w, steps = 1080, 1920, 128
q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device="cuda")
out = {}
for z in (1.0, -3.0):
    t = torch.tensor([[0.0, 0.0, z]], device="cuda")
    for mode in (4, 1):
        with torch.no_grad():
            a = H.make_loop(H.spec_to_module(O.scene_test2()), h, w)(q, t, mode, 1, steps)
            b = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, early_out=False)(q, t, mode, 1, steps)   # no early-out: no parking
        assert torch.equal(a, b), (z, mode)
print("PARKING-OK")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # the variant libraries go to a directory of their own (RM_LIB_DIR): the product library is never replaced
    env = dict(os.environ, RM_HIPCC_EXTRA="-DRM_PARKING", RM_PARK="1", RM_SPECIALIZE="jit", RM_LIB_DIR=str(tmp_path / "lib"))
    from ray_marching_amd import _build
    before = open(_build.LIB_PATH + ".srchash").read()
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert open(_build.LIB_PATH + ".srchash").read() == before and os.path.isfile(tmp_path / "lib" / "librm_hip.so")
    assert r.returncode == 0 and "PARKING-OK" in r.stdout, r.stderr[-2000:]


def _hip_band_worker(rank, world, port, h, w, steps, out_dir):
    """One rank of a row-tiled render with the real HIP band renderer (both ranks share the box's one GPU; gloo
    carries the collectives, device tensors staged through the host)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from ray_marching_amd.control import RenderLoop
        from ray_marching_amd.distributed import RowTileRenderer, all_reduce_gradients, row_band
        from ray_marching_amd.scene.scene_registry import make_closed_test_scene, make_test_scene2
        torch.cuda.set_device(0)
        band = row_band(h, rank, world)
        loop = RenderLoop(make_test_scene2(), num_cameras=1, px_width=w, px_height=h, focal_length=H.PX * h,
                          sensor_width=H.PX * w, sensor_height=H.PX * h, normals_eps=H.EPS, rows=band).to(DEV)
        assert loop.camera.ray_positions.shape[1] == band[1] - band[0]          # only this rank's band is resident
        q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -3.0]], device=DEV)
        frames = {}
        for exchange in ("p2p", "gather"):
            tiles = RowTileRenderer(loop, exchange=exchange)
            for mode in (4, 0, 1, 5):
                with torch.no_grad():
                    frame = tiles.render(q, t, mode, 1, steps, dst=0)
                assert (frame is not None) == (rank == 0)
                if rank == 0:
                    frames[(exchange, mode)] = frame.cpu()
        # training: each rank back-propagates the loss of its own band, gradients summed over the ranks
        scene = make_closed_test_scene()
        tl = RenderLoop(scene, num_cameras=1, px_width=w, px_height=h, focal_length=H.PX * h, sensor_width=H.PX * w,
                        sensor_height=H.PX * h, normals_eps=H.EPS, rows=band).to(DEV)
        tl(q, torch.tensor([[0.0, 0.0, -1.0]], device=DEV), 0, 1, 24).pow(2).sum().backward()
        cpu_holder = torch.nn.ParameterList([torch.nn.Parameter(p.detach().cpu()) for p in scene.parameters()])
        for hp, p in zip(cpu_holder, scene.parameters()):      # gloo: reduce on the host
            hp.grad = p.grad.cpu()
        all_reduce_gradients(cpu_holder)
        if rank == 0:
            torch.save({"frames": frames, "grads": [p.grad for p in cpu_holder]}, os.path.join(out_dir, "rank0.pt"))
    finally:
        dist.destroy_process_group()


def test_hip_band_renderer_under_a_real_process_group(tmp_path):
    """RowTileRenderer with the HIP RenderLoop as band renderer, two processes in one torch.distributed group
    (gloo; both on this box's single GPU): frames equal the single-process render bit for bit -- per-pixel shaders
    and, after the min/max all-reduce, the globally normalised ones; the point-to-point exchange equals
    dist.gather; summed band gradients equal the whole-frame gradients."""
    import socket
    import torch.multiprocessing as mp
    from ray_marching_amd.scene.scene_registry import make_closed_test_scene
    h, w, steps, world = 90, 160, 48, 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_hip_band_worker, args=(world, port, h, w, steps, str(tmp_path)), nprocs=world, join=True)
    got = torch.load(os.path.join(tmp_path, "rank0.pt"), weights_only=False)      # written by this test
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -3.0]], device=DEV)
    for (exchange, mode), frame in got["frames"].items():
        with torch.no_grad():
            want = loop(q, t, mode, 1, steps).cpu()
        assert torch.equal(frame, want), (exchange, mode)
    scene = make_closed_test_scene()
    whole = H.make_loop(scene, h, w)
    whole(q, torch.tensor([[0.0, 0.0, -1.0]], device=DEV), 0, 1, 24).pow(2).sum().backward()
    for g, p in zip(got["grads"], scene.parameters()):
        assert (g - p.grad.cpu()).abs().max().item() <= 2e-5 * max(1.0, p.grad.abs().max().item())


def test_fp16_standalone_modules_equal_fp32_arithmetic_on_fp16_storage():
    """rm_camera_forward / rm_march_forward / rm_normals_forward / rm_sdf_forward with dtype = F16: the
    stand-alone modules of a .half() scene read and write fp16 arrays directly; results = the fp32 oracle on the
    fp16-rounded inputs, rounded once to fp16, bit for bit."""
    from ray_marching_amd.rendering.ray_marching import PinholeCamera, SDFMarcher, SDFNormals
    h, w, steps = 36, 52, 40
    r = lambda x: x.half().float()
    spec = O.scene_test2()
    spec16 = O.map_spec(spec, r)
    module = H.spec_to_module(spec).to(DEV).half()
    cam = PinholeCamera(2, w, h, H.PX * h, H.PX * w, H.PX * h).to(DEV).half()
    gen = torch.Generator().manual_seed(2)
    q = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0, 0.0, 0.0]]) + 0.1 * torch.randn(2, 4, generator=gen), dim=-1).half()
    t = (torch.tensor([[0.0, 0.0, -3.0]]) + 0.2 * torch.randn(2, 3, generator=gen)).half()
    bufs = tuple(r(b) for b in O.camera_buffers(2, w, h, H.PX * h, H.PX * w, H.PX * h))
    with torch.no_grad():
        pos, frames, _, dirs = cam(q.to(DEV), t.to(DEV))
        want_pos, want_frames, want_dirs = O.camera_forward(*bufs, q.float(), t.float())
        assert pos.dtype == dirs.dtype == frames.dtype == torch.float16
        assert torch.equal(pos.cpu(), want_pos.half()) and torch.equal(dirs.cpu(), want_dirs.half())
        assert torch.equal(frames.cpu(), want_frames.half())
        p = SDFMarcher(module)(pos, dirs, steps)                    # fp16 in, fp32 march, fp16 out
        want_p = O.march(spec16, pos.float().cpu(), dirs.float().cpu(), steps).half()
        assert p.dtype == torch.float16 and torch.equal(p.cpu(), want_p)
        nrm = SDFNormals(module, H.EPS).to(DEV).half()
        n, lap = nrm(p)
        tetra = tuple(r(c) for c in O.tetra_constants(H.EPS))
        want_n, want_lap = O.normals(spec16, p.float().cpu(), H.EPS, tetra)
        assert n.dtype == lap.dtype == torch.float16
        assert H.report("fp16 normals", n, want_n.half())[0] == 0.0          # report(): identical NaN patterns too
        assert H.report("fp16 laplacian", lap, want_lap.half())[0] == 0.0
        d = module(p)
        assert d.dtype == torch.float16 and torch.equal(d.cpu(), O.sdf_eval(spec16, p.float().cpu()).half())
