"""Ray regeneration (RM_FLAG_REGEN, include/rm_abi.h): the pool kernels must render the SAME image as the tile
kernel, bit for bit, whatever the dealing order -- and the tile kernel is what tests/test_gpu_parity.py pins to
the oracle.  One direct oracle comparison is kept here as well."""
import ctypes as C

import pytest
import torch

from oracle import sdf_oracle as O
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _same_bits(a, b):
    a, b = a.contiguous(), b.contiguous()
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    iv = {2: torch.int16, 4: torch.int32, 8: torch.int64}[a.element_size()]
    return bool(torch.equal(a.view(iv), b.view(iv)))


def _poses(n, z, seed):
    gen = torch.Generator().manual_seed(seed)
    q = torch.nn.functional.normalize(torch.tensor([[1.0, 0.0, 0.0, 0.0]]) + 0.1 * torch.randn(n, 4, generator=gen), dim=-1)
    t = torch.tensor([[0.0, 0.0, z]]) + 0.2 * torch.randn(n, 3, generator=gen)
    return q.to(DEV), t.to(DEV)


@pytest.mark.parametrize("per_ray", [False, True])
def test_regen_equals_tile_kernel_every_mode_orders_and_overhanging_tiles(per_ray):
    """Two cameras, a frame whose size is no multiple of the 8x8 tile, all eight shaders; frame 1 is dealt in natural
    order and records the ray costs, frames 2 and 3 are dealt by the tile scores / per-ray order made of them."""
    n, h, w, steps = 2, 516, 523, 48
    tile = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, n=n, regen=False, adaptive_order=0)
    pool = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, n=n, regen=True, order_per_ray=per_ray, adaptive_order=2)
    for z in (-3.0, 1.0):
        q, t = _poses(n, z, 3)
        for mode in range(8):
            if mode == 3:
                continue            # the reference's vignette shader only broadcasts for one camera (shader.py:64)
            with torch.no_grad():
                want = tile(q, t, mode, 2, steps)
                for frame in range(3):
                    got = pool(q, t, mode, 2, steps)
                    assert _same_bits(got, want), (z, mode, frame)
    st = [s for s in pool._order_state.values() if s.get("T")]
    assert st and all(s["valid"] for s in st) and st[0]["order"].numel() == (st[0]["T"] * (64 if per_ray else 1))
    # the order in use is a permutation
    o = st[0]["order"].long()
    assert int(o.min()) == 0 and int(o.max()) == o.numel() - 1 and o.unique().numel() == o.numel()


def test_regen_against_the_oracle():
    n, h, w, steps = 1, 40, 56, 32
    spec = O.scene_test2()
    loop = H.make_loop(H.spec_to_module(spec), h, w, n=n, regen=True)
    bufs = O.camera_buffers(n, w, h, H.PX * h, H.PX * w, H.PX * h)
    q, t = _poses(n, 1.0, 11)
    for mode in (0, 1, 4, 5, 6):
        with torch.no_grad(), O.math_mode("restated"):
            want = O.render(spec, bufs, q.cpu(), t.cpu(), mode, 2, steps, H.EPS, cmap=loop.shader.cyclic_cmap.cpu())
            got = loop(q, t, mode, 2, steps)
        worst, _ = H.report(f"regen mode {mode}", got, want)
        assert worst <= (0.0 if mode != 6 else 1.2e-7), (mode, worst)


def test_regen_fp16_module_row_band_and_fallbacks():
    n, h, w = 1, 300, 410
    spec = O.scene_test2()
    tile = H.make_loop(H.spec_to_module(spec), h, w, n=n, regen=False, adaptive_order=0).to(torch.float16)
    pool = H.make_loop(H.spec_to_module(spec), h, w, n=n, regen=True).to(torch.float16)
    q, t = _poses(n, 1.0, 5)
    with torch.no_grad():
        for mode in (0, 2, 7):
            assert _same_bits(pool(q.half(), t.half(), mode, 1, 64), tile(q.half(), t.half(), mode, 1, 64)), mode
        # a row band, as a rank of a row-tiled render asks for
        assert _same_bits(pool(q.half(), t.half(), 4, 1, 64, rows=(37, 203)), tile(q.half(), t.half(), 4, 1, 64, rows=(37, 203)))
        # step counts the pools do not take (not a multiple of 4) fall back to the tile kernel
        assert _same_bits(pool(q.half(), t.half(), 4, 1, 30), tile(q.half(), t.half(), 4, 1, 30))
        assert _same_bits(pool(q.half(), t.half(), 4, 1, 0), tile(q.half(), t.half(), 4, 1, 0))
    # a training frame records a trajectory: tile kernel, gradients as ever
    pool32 = H.make_loop(H.spec_to_module(spec), 64, 64, regen=True)
    img = pool32(*_poses(1, -3.0, 1), 0, 1, 32)
    img.mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in pool32.scene.parameters())


def test_regen_flag_is_refused_without_its_buffers():
    from ray_marching_amd import _abi, ops
    from ray_marching_amd.compiler import compiled_for
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), 64, 64)
    cs = compiled_for(loop.scene)
    rp, rd = loop.camera.ray_positions, loop.camera.ray_directions
    q, t = _poses(1, -3.0, 1)
    s, keep = cs.scene_struct(None, rp.device)
    cam = ops.camera_struct(rp, rd)
    image = torch.empty(1, 64, 64, 3, device=DEV)
    work = torch.empty(_abi.WORK_WORDS, dtype=torch.int32, device=DEV)
    flags = ops.default_flags(True, True, True, True)
    rc = _abi.lib.rm_render_forward(s, cam, loop.normals.tetra(), _abi.ptr(q), _abi.ptr(t), _abi.ptr(image), _abi.dtype_code(torch.float32),
                                    None, None, None, None, None, _abi.ptr(work), None, 0, 0, 4, 1, 32, 0, 64, flags, None, None, None, 0,
                                    _abi.current_stream(rp.device))
    assert rc == -1 and b"RM_FLAG_REGEN" in _abi.lib.rm_last_error()
    torch.cuda.synchronize()


def test_order_sort_and_tile_scores_against_torch():
    """rm_tile_order_from_cost, one-block and multi-block path: the stable sort by descending cost class;
    rm_tile_score_from_ray_cost: classes 16..31 by the number of long rays, 0..15 by the longest ray."""
    from ray_marching_amd import _abi
    gen = torch.Generator().manual_seed(7)
    stream = _abi.current_stream(torch.device(DEV))
    for n, top in ((5000, 128), (131072, 31), (131073, 128), (2_073_600, 128), (777_777, 255)):
        cost = torch.randint(0, top + 1, (n,), generator=gen, dtype=torch.int32)
        cost[torch.rand(n, generator=gen) < 0.3] = top                    # a big class, like the rays that never settle
        cost_d = cost.to(DEV)
        order = torch.full((n,), -1, dtype=torch.int32, device=DEV)
        scratch = torch.empty(_abi.ORDER_SCRATCH_INTS, dtype=torch.int32, device=DEV)
        _abi.check(_abi.lib.rm_tile_order_from_cost(_abi.ptr(cost_d), n, top, _abi.ptr(order), _abi.ptr(scratch), stream), "order")
        cls = 31 - (cost.long() * 32) // (top + 1)
        want = torch.sort(cls, stable=True).indices
        assert torch.equal(order.cpu().long(), want), (n, top)
    # more items than one block sorts, no scratch buffer: refused
    rc = _abi.lib.rm_tile_order_from_cost(_abi.ptr(cost_d), 777_777, 255, _abi.ptr(order), None, stream)
    assert rc == -1 and b"scratch" in _abi.lib.rm_last_error()
    cams, tx, ty, S = 2, 50, 30, 128
    T = cams * tx * ty
    ray = torch.randint(0, 60, (T, 64), generator=gen, dtype=torch.int32)
    for i in range(0, T, 7):                                               # every seventh tile gets some long rays
        k = int(torch.randint(1, 65, (1,), generator=gen))
        ray[i, :k] = torch.randint(96, S + 1, (k,), generator=gen, dtype=torch.int32)
    ray_d = ray.to(DEV)
    for reach in (0, 1, 2):
        score = torch.empty(T, dtype=torch.int32, device=DEV)
        raw = torch.empty(T, dtype=torch.int32, device=DEV)
        _abi.check(_abi.lib.rm_tile_score_from_ray_cost(_abi.ptr(ray_d), T, tx, ty, reach, S, _abi.ptr(raw), _abi.ptr(score), stream), "score")
        n_long = (ray >= 96).sum(dim=1).view(cams, 1, ty, tx).float()
        mx = ray.max(dim=1).values.view(cams, 1, ty, tx).float()
        k = 2 * reach + 1
        near_long = torch.nn.functional.max_pool2d(n_long, k, 1, reach).long().flatten()
        near_max = torch.nn.functional.max_pool2d(mx, k, 1, reach).long().flatten()
        own = n_long.long().flatten()
        want = torch.where(own > 0, 17 + ((own - 1) * 15) // 64, torch.where(near_long > 0, torch.full_like(own, 16), (near_max * 16) // (S + 1)))
        assert torch.equal(score.cpu().long(), want), reach


def test_captured_frame_with_regen_follows_the_camera():
    """regen=True under HIP-graph replay: two graphs (plain / recording + order renewal) share the static buffers."""
    n, h, w, steps = 1, 520, 528, 32
    tile = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=False, adaptive_order=0)
    pool = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=True, adaptive_order=3)
    frame = pool.capture(4, 1, steps)
    assert frame.graph_record is not None
    for i in range(8):
        q, t = _poses(n, 1.0 - 0.5 * i, 20 + i)
        with torch.no_grad():
            assert _same_bits(frame(q, t), tile(q, t, 4, 1, steps)), i


def test_auto_mode_probes_both_kernels_and_never_changes_a_pixel():
    """regen="auto" (the default) at 1080p: whichever kernel the timing heuristic is on, every frame has the tile
    kernel's bits, and both kernels were really probed.  WHICH kernel wins on a shared box is printed, not asserted
    (profiles/regen_probe.py: the pools inside the torus (0,0,1), the tile kernel in front of the scene (0,0,-3),
    margins of 13-27 % that one noisy probe can flip); the decision logic itself is tested with injected timings in
    tests/test_host_cpu.py::test_auto_kernel_choice_state_machine."""
    h, w, steps = 1080, 1920, 128
    tile = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=False, adaptive_order=0)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV)
    for z, want_regen in ((1.0, True), (-3.0, False)):
        auto = H.make_loop(H.spec_to_module(O.scene_test2()), h, w)
        assert auto.regen == "auto"
        t = torch.tensor([[0.0, 0.0, z]], device=DEV)
        with torch.no_grad():
            want = tile(q, t, 4, 1, steps)
            for i in range(160):                  # every frame waited for, like an interactive loop: a single noisy
                got = auto(q, t, 4, 1, steps)     # measurement is corrected by the next probe well within these frames
                if i % 16 < 3:
                    assert _same_bits(got, want), (z, i)
                torch.cuda.synchronize()
        (st,) = auto._choice_state.values()
        assert st["log"], "no pair of probes completed in 160 waited-for frames"
        for n_frame, was_regen, ms_other, ms_used in st["log"]:
            assert 0.0 < ms_other < 1e3 and 0.0 < ms_used < 1e3, st["log"]
        print(f"regen=auto at z={z:+g}: ends on {'the pool kernels' if st['regen'] else 'the tile kernel'} "
              f"(expected on a quiet box: {'pools' if want_regen else 'tile'}); probes (frame, pools in use, ms other, ms used): {st['log'][-4:]}")


@pytest.mark.parametrize("seed", range(8))
def test_regen_random_scene_trees_equal_tile_kernel(seed):
    """Random compositions of all 11 node types through the interpreter: pool kernels == tile kernel, bit for bit
    (NaN pixels included), at a pose inside and a pose outside the geometry, for a shader of each kind."""
    gen = torch.Generator().manual_seed(4000 + seed)
    spec = H.random_spec(gen)
    n, h, w, steps = 1, 268, 284, 40
    tile = H.make_loop(H.spec_to_module(spec), h, w, n=n, regen=False, adaptive_order=0)
    pool = H.make_loop(H.spec_to_module(spec), h, w, n=n, regen=True, adaptive_order=2)
    for z in (-4.0, 0.3):
        q, t = _poses(n, z, 50 + seed)
        for mode in (0, 2, 5, 7):
            with torch.no_grad():
                want = tile(q, t, mode, 3, steps)
                for frame in range(2):
                    assert _same_bits(pool(q, t, mode, 3, steps), want), (seed, z, mode, frame)


@pytest.mark.parametrize("z", [-3.0, 1.0])
def test_regen_config3_full_size_fp16(z):
    """BASELINE configs[2] (3840x2160, 256 steps, float16 module): pool kernels with natural, tile-score and per-ray
    dealing orders == tile kernel, bit for bit."""
    h, w, steps = 2160, 3840, 256
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV).half()
    t = torch.tensor([[0.0, 0.0, z]], device=DEV).half()
    tile = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=False, adaptive_order=0).to(torch.float16)
    with torch.no_grad():
        want = {m: tile(q, t, m, 1, steps) for m in (4, 0)}
    del tile
    for per_ray in (False, True):
        pool = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=True, order_per_ray=per_ray, adaptive_order=2).to(torch.float16)
        with torch.no_grad():
            for frame in range(3):
                for m in (4, 0):
                    assert _same_bits(pool(q, t, m, 1, steps), want[m]), (z, per_ray, frame, m)
        del pool


def test_regen_config5_band_full_size():
    """BASELINE configs[4]: rows 1620..2160 of the 7680x4320 frame of the 32-primitive scene, 256 steps."""
    from ray_marching_amd.scene.scene_registry import make_many_primitive_scene
    h, w, steps = 4320, 7680, 256
    band = (1620, 2160)
    q = torch.tensor([[1.0, 0.0, 0.0, 0.0]], device=DEV); t = torch.tensor([[0.0, 0.0, -4.5]], device=DEV)
    tile = H.make_loop(make_many_primitive_scene(32), h, w, regen=False, rows=band)
    pool = H.make_loop(make_many_primitive_scene(32), h, w, regen=True, rows=band, adaptive_order=2)
    with torch.no_grad():
        want = tile(q, t, 4, 1, steps, rows=band)
        for frame in range(3):
            assert _same_bits(pool(q, t, 4, 1, steps, rows=band), want), frame


def test_regen_with_the_minmax_hook_of_a_row_tiled_render():
    """Globally normalised shaders over two row bands, min/max folded across the launches through the
    allreduce_minmax hook (what RowTileRenderer does across ranks): pools == tile kernel == the unsplit frame."""
    n, h, w, steps = 1, 272, 296, 32
    q, t = _poses(n, 1.0, 9)
    tile = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=False, adaptive_order=0)
    pool = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=True)
    for mode in (1, 2, 5):
        with torch.no_grad():
            whole = tile(q, t, mode, 1, steps)
            for loop in (tile, pool):
                # pass 1: every band reports its min/max; pass 2: every band normalises with the global pair
                seen = []
                for band in ((0, 100), (100, h)):
                    loop(q, t, mode, 1, steps, rows=band, allreduce_minmax=lambda lohi: seen.append(lohi.clone()))
                lo = torch.stack([s[0] for s in seen]).min()
                hi = torch.stack([s[1] for s in seen]).max()

                def globalise(lohi):
                    lohi[0], lohi[1] = lo, hi
                parts = [loop(q, t, mode, 1, steps, rows=band, allreduce_minmax=globalise) for band in ((0, 100), (100, h))]
                assert _same_bits(torch.cat(parts, dim=1), whole), (mode, loop is pool)


def test_auto_mode_inside_a_user_graph_capture():
    """A default RenderLoop (regen="auto") at a size where auto is active, captured by the user with torch.cuda.graph:
    no timing events, no kernel choice inside the capture; replays render the frame."""
    h, w, steps = 1080, 1920, 32
    loop = H.make_loop(H.spec_to_module(O.scene_test2()), h, w)
    q, t = _poses(1, -3.0, 2)
    with torch.no_grad():
        want = loop(q, t, 4, 1, steps).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            loop(q, t, 4, 1, steps)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            img = loop(q, t, 4, 1, steps)
        g.replay()
        torch.cuda.synchronize()
    assert _same_bits(img, want)


def test_regen_in_a_band_only_loop():
    """RenderLoop(rows=band) -- what a rank of a row-tiled render builds: it holds only its band of the camera
    buffers -- with the pool kernels == the same rows of the whole frame from the tile kernel."""
    n, h, w, steps = 1, 520, 536, 32
    band = (136, 424)
    q, t = _poses(n, 1.0, 13)
    whole = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=False, adaptive_order=0)
    pool = H.make_loop(H.spec_to_module(O.scene_test2()), h, w, regen=True, adaptive_order=2, rows=band)
    assert pool.camera.ray_positions.shape[1] == band[1] - band[0]
    for mode in (4, 1, 7):
        with torch.no_grad():
            want = whole(q, t, mode, 1, steps, rows=band) if mode == 1 else whole(q, t, mode, 1, steps)[:, band[0]:band[1]]
            for frame in range(3):
                assert _same_bits(pool(q, t, mode, 1, steps), want), (mode, frame)
