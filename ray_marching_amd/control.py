"""RenderLoop with the call surface of the reference's control.py:197-258.

The reference chains camera -> marcher -> scene -> normals -> shader as separate module calls
(hundreds of ATen launches per frame).  Here ``forward`` issues ONE fused HIP kernel
(rm_render_forward: k_render_fwd) plus, for shader modes 1/2/5, the tiny normalisation pass.
The sub-modules are still constructed (same attribute names, same buffers) so code that
reaches into ``render_loop.camera`` etc. keeps working.  Nothing here imports pynput /
pyautogui (the reference's control.py does, at import time: control.py:3-4,14).
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
from torch import Tensor

from . import _abi, ops
from .compiler import compiled_for
from .rendering.ray_marching import PinholeCamera, SDFMarcher, SDFNormals
from .rendering.shader import Shader


_AB_PACK = bool(os.environ.get("RM_AB_PACK"))      # experiment knob: pack the parameters for inference frames too


class CapturedFrame:
    """One inference frame of a RenderLoop captured into a HIP graph (torch.cuda.CUDAGraph).

    A frame is 2-3 kernel launches (workspace init, k_render_fwd, optional k_shade_finish) plus some tens of
    microseconds of Python; for small frames (main.py's 1440x900x32 takes ~0.15 ms of GPU time) the host side
    matters.  Replaying a captured graph removes it.  The pose lives in static device buffers refreshed (one tiny
    copy each) before every replay; the scene parameters are gathered by the kernel from the nn.Parameter
    storages themselves (CompiledScene.param_table), so edits and optimiser steps need nothing at all, and a
    re-allocated parameter only a refresh of the pointer table.  Changing mode, steps, resolution or the scene
    topology needs a new capture.  Inference only (no autograd)."""

    def __init__(self, loop: "RenderLoop", mode: int = 0, degree: int = 1, marching_steps: int = 32, rows=None,
                 display: bool = False):
        """``display=True``: the replay writes main.py:78-84's display tensor ([H,W,4] fp32, alpha 1: RenderLoop.display_frame)
        instead of the [N,H,W,3] image (one camera)."""
        self.loop, self.mode, self.degree, self.steps, self.rows = loop, mode % 8, int(degree), int(marching_steps), rows
        self.display = bool(display)
        rp, rd = loop._io_buffers(False)
        dev, n = rp.device, rp.shape[0]
        self.cs = compiled_for(loop.scene)
        self.q = torch.zeros(n, 4, dtype=rp.dtype, device=dev)
        self.q[:, 0] = 1.0
        self.t = torch.zeros(n, 3, dtype=rp.dtype, device=dev)
        self._table_src = self.cs.param_table(dev)
        self.table = None if self._table_src is None else self._table_src.clone()
        with torch.no_grad():
            self.params = None if self.table is not None else self.cs.pack_params(dev).clone()
        cmap = loop._cmap(dev) if self.mode in (6, 7) else None
        flags = ops.default_flags(loop.early_out, loop.tile8x8, loop.dynamic_tiles, loop.regen is True, loop.order_per_ray)
        cs, table, params = self.cs, self.table, self.params

        class _Static:          # a CompiledScene view whose scene_struct points at THIS object's static buffers
            def __getattr__(self, name):
                return getattr(cs, name)

            def scene_struct(self, prm, device, table_=None, **block):
                return cs.scene_struct(params, device, table, **block)

        static = _Static()

        def frame(order=None, cost=None):
            return ops.render_frame(None, self.q, self.t, static, rp, rd, loop.normals.tetra(), cmap,
                                    self.mode, self.degree, self.steps, rows, flags, None, loop.precision,
                                    "rgba" if self.display else None, order, cost)

        # regen=True: the regeneration kernels need a dealing order that follows the camera.  Two graphs share the
        # static buffers: the plain frame, and a frame that also records the ray costs and renews the order behind it;
        # every `adaptive_order`-th replay is of the second kind.  ("auto" needs host decisions between frames and
        # stays with the tile kernel under capture.)
        self._calls, self.graph_record, self._order = 0, None, None
        if ops.regen_applies(flags, self.steps, False) and loop.adaptive_order > 0:
            st = loop._new_order_state(rp, rows, True)
            self._order = st if st["T"] else None
        self._period = loop.adaptive_order
        st = self._order
        with torch.no_grad():
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    frame()                       # warm-up outside capture (library / allocator state)
                if st is not None:                # first order (at the pose in the static buffers)
                    frame(None, st["cost"])
                    loop._renew_order(st, self.steps, dev)
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.image = frame(None if st is None else st["order"])
            if st is not None:
                self.graph_record = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_record, pool=self.graph.pool()):
                    img = frame(st["order"], st["cost"])
                    loop._renew_order(st, self.steps, dev)
                    self.image.copy_(img)

    def __call__(self, orientations: Tensor, translations: Tensor) -> Tensor:
        """Render with the given pose; returns the graph's static output tensor [N,rows,W,3]
        (overwritten by the next replay -- clone it to keep it)."""
        with torch.no_grad():
            self.q.copy_(orientations)
            self.t.copy_(translations)
            if self.table is not None:
                cur = self.cs.param_table(self.table.device)
                if cur is None:
                    raise RuntimeError("CapturedFrame: a scene parameter can no longer be read in place "
                                       "(moved off the device / changed dtype): capture again")
                if cur is not self._table_src:    # a parameter was re-allocated: same layout, new pointers
                    self.table.copy_(cur)
                    self._table_src = cur
            else:
                self.params.copy_(self.cs.pack_params(self.params.device))
        self._calls += 1
        if self.graph_record is not None and self._calls % self._period == 0:
            self.graph_record.replay()
        else:
            self.graph.replay()
        return self.image


class RenderLoop(nn.Module):
    def __init__(self, scene, num_cameras: int = 1, px_width: int = 800, px_height: int = 800,
                 focal_length: float = 17e-3, sensor_width: float = 17e-3, sensor_height: float = 17e-3,
                 normals_eps: float = 5e-2, early_out: bool = True, tile8x8: bool = True,
                 dynamic_tiles: bool = True, precision: str = "exact", rows=None, adaptive_order: int = 16,
                 regen="auto", order_per_ray: bool = False):
        """Arguments of the reference's RenderLoop (control.py:198-208) plus kernel options.  ``rows=(r0, r1)``:
        this loop only ever renders that band of the frame (one rank of a row-tiled multi-GPU render) and keeps
        only that band of the camera buffers; ``forward`` then returns [N, r1-r0, W, 3]."""
        super().__init__()
        self.scene = scene
        self.px_width = px_width
        self.px_height = px_height
        self.camera = PinholeCamera(num_cameras=num_cameras, px_width=px_width, px_height=px_height,
                                    focal_length=focal_length, sensor_width=sensor_width,
                                    sensor_height=sensor_height, rows=rows)
        self.marcher = SDFMarcher(sdf_scene=self.scene, early_out=early_out)
        self.normals = SDFNormals(sdf_scene=self.scene, normals_eps=normals_eps)
        self.shader = Shader()
        self.early_out = early_out
        self.tile8x8 = tile8x8
        self.dynamic_tiles = dynamic_tiles
        if precision not in ("exact", "fast"):
            raise ValueError("precision must be 'exact' (bit-faithful to the reference's CPU op stream, default) or 'fast'")
        self.precision = precision
        # frames deal their wave tiles longest-first, from the step counts an earlier frame recorded; the
        # order is renewed every `adaptive_order` frames (0 = natural order always).  State per (band, step count,
        # stream): two streams must never share an order buffer that one of them is rewriting.
        self.adaptive_order = int(os.environ.get("RM_ADAPTIVE_ORDER", adaptive_order))      # env: A/B probes
        self._order_state = {}
        self._f32_cache = {}
        # Ray regeneration (RM_FLAG_REGEN): lanes whose ray is done take the next ray of a queue instead of waiting
        # for the slowest ray of their tile -- for poses where few rays per tile settle late.  True / False, or
        # "auto": large inference frames time both kernels on the launch stream now and then and use the faster one
        # (_choose_kernel).
        regen = os.environ.get("RM_REGEN", regen)
        if regen in ("0", "1"):
            regen = bool(int(regen))
        if regen not in (True, False, "auto"):
            raise ValueError("regen must be True, False or 'auto'")
        self.regen = regen
        self._choice_state = {}
        # its dealing order: per tile (how many of a tile's rays march long -- stable under camera motion), or per ray
        # (which rays do: only right for a frame that is rendered again unchanged, a paused viewer)
        self.order_per_ray = bool(int(os.environ.get("RM_ORDER_PER_RAY", int(order_per_ray))))

    def _apply(self, fn, *args, **kwargs):
        """.to(device) / .half() / ...: buffers keyed on the old placement (dealing orders, cost maps, kernel choice,
        converted camera buffers) are dropped rather than carried to a device they do not live on."""
        self._order_state, self._choice_state, self._f32_cache = {}, {}, {}
        return super()._apply(fn, *args, **kwargs)

    # cached conversions live outside the module's picklable state (copy.deepcopy / torch.save of a RenderLoop)
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_f32_cache"] = {}
        state["_order_state"] = {}
        state["_choice_state"] = {}
        return state

    def _f32_buffer(self, name: str) -> Tensor:
        """Camera buffer as contiguous fp32 (training frames: the backward kernels are fp32).  When the module
        was cast with .to(float16) the values are the fp16-rounded ones, like the reference's cast buffers."""
        buf = getattr(self.camera, name)
        if buf.dtype == torch.float32 and buf.is_contiguous():
            return buf
        key = (name, buf.data_ptr(), buf._version, buf.dtype)
        hit = self._f32_cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, buf.float().contiguous())
            self._f32_cache[name] = hit
        return hit[1]

    def _io_buffers(self, training: bool):
        """The camera buffers the frame kernel reads: as they are (fp32 or fp16 -- no cast pass) for inference,
        fp32 copies for training frames or exotic module dtypes."""
        rp, rd = self.camera.ray_positions, self.camera.ray_directions
        if not training and rp.dtype in (torch.float32, torch.float16) and rp.dtype == rd.dtype \
                and rp.is_contiguous() and rd.is_contiguous():
            return rp, rd
        return self._f32_buffer("ray_positions"), self._f32_buffer("ray_directions")

    def _cmap(self, device) -> Tensor:
        """The colormap as the kernel reads it: the registered buffer itself (float64 as loaded, or whatever
        .to(dtype) made of it) when it is on the device, else a cached device copy."""
        cm = self.shader.cyclic_cmap
        ok = cm.dtype in (torch.float64, torch.float32, torch.float16)
        if ok and cm.device == torch.device(device) and cm.is_contiguous():
            return cm
        key = (cm.data_ptr(), cm._version, cm.dtype, str(device))
        hit = self._f32_cache.get("cmap")
        if hit is None or hit[0] != key:
            hit = (key, cm.to(device=device, dtype=cm.dtype if ok else torch.float32).contiguous())
            self._f32_cache["cmap"] = hit
        return hit[1]

    # frames smaller than this never take the regeneration kernels in "auto" mode: a 1080p frame is 6 draws per pool
    # lane, below ~1 M rays the pools hardly refill at all
    REGEN_AUTO_MIN_RAYS = 1 << 20
    # tiles this close to a tile with long rays are dealt right after those (rm_tile_score_from_ray_cost): covers a
    # camera that moves by up to 8 pixels per tile of reach before the order is renewed
    ORDER_REACH_TILES = int(os.environ.get("RM_ORDER_REACH", 2))

    def _choose_kernel(self, rp: Tensor, rows, steps: int):
        """regen="auto": (use the regeneration kernels for this frame?, record the dealing order now?, list for the
        frame's timing events or None).  Frames come in cycles of `adaptive_order` (16 if that is off): the first one
        of a cycle runs the kernel NOT in use (and renews its dealing order, so that it is measured with an order as old
        as it would be in use), the second one the kernel in use, both between timing events on the launch stream that
        are looked at -- without waiting -- by later frames.  The pools take over when they were more than 3 % faster
        and are then checked against the tile kernel every cycle (it takes over again as soon as it is faster at all);
        while the tile kernel is in use the pools are looked at every fourth cycle: such a probe costs their time plus
        the order renewal, ~85 us at 1080p in front of the scene, 0.6 % of the headline.  The choice follows the GPU
        only as closely as the host does: a loop that enqueues hundreds of frames without ever waiting decides for
        frames the GPU renders much later."""
        n, h, w, _ = rp.shape
        r0, r1 = rows if rows is not None else (0, h)
        return self._choose_kernel_for((r0, r1, steps, rp.device.index, torch.cuda.current_stream(rp.device).cuda_stream))

    def _choose_kernel_for(self, key):
        """The state machine of _choose_kernel for one (band, step count, stream)."""
        st = self._choice_state.get(key)
        if st is None:
            st = self._choice_state[key] = {"regen": False, "n": 0, "pending": {}, "ms": {}, "skip": 0, "log": []}
        for name in list(st["pending"]):
            pair = st["pending"][name]
            if not pair:                          # the frame that was to be timed never recorded its events (it raised)
                st["pending"], st["ms"], st["await_used"] = {}, {}, False
                break
            if pair[0][1].query():
                st["ms"][name] = pair[0][0].elapsed_time(pair[0][1])
                del st["pending"][name]
        if len(st["ms"]) == 2:
            st["log"].append((st["n"], st["regen"], round(st["ms"]["other"], 4), round(st["ms"]["used"], 4)))   # diagnostics
            del st["log"][:-64]
            # Asymmetric on purpose (profiles/orbit_probe.py, a camera flying through the scene): the pools in the wrong
            # place cost +25 %, the tile kernel in the wrong place misses 12-16 %; a probe of the tile kernel is free
            # where it is the faster one, a probe of the pools costs their time plus an order renewal.
            if st["regen"]:
                # the tile kernel takes over at once when it is clearly faster (> 10 %), and on the second probe in a row
                # that says so by less: one noisy measurement of a pool frame must not cost four cycles on the slower
                # kernel (a shared box: bench `poses` at (0,0,1) 0.379 ms under auto against 0.325 ms for the pools alone)
                if st["ms"]["other"] < 0.9 * st["ms"]["used"] or (st["ms"]["other"] < st["ms"]["used"] and st.get("doubt")):
                    st["regen"] = False
                    st["doubt"] = False
                else:
                    st["doubt"] = st["ms"]["other"] < st["ms"]["used"]
                # while the pools are in use (or were a moment ago) the tile kernel is looked at every cycle, every
                # second one where it is far behind
                st["skip"] = 1 if st["ms"]["other"] > 1.15 * st["ms"]["used"] else 0
            elif st["ms"]["other"] < 0.97 * st["ms"]["used"]:
                st["regen"] = True
                st["skip"] = 0
            elif not st.get("warm"):
                # the very first look at the pools ran them cold -- no dealing order yet (natural order: ~12 % slower),
                # first-use allocations: 0.505 ms against 0.32 ms two probes later at (0,0,1) -- so it only counts when
                # they win; they are looked at again in the next cycle before the every-fourth-cycle rhythm starts
                st["skip"] = 0
            else:
                st["skip"] = 3          # the tile kernel stays: the pools are looked at every fourth cycle
            st["warm"] = True
            st["ms"] = {}
        cycle = self.adaptive_order if self.adaptive_order > 0 else 16
        phase = st["n"] % cycle
        st["n"] += 1
        if cycle > 1 and phase == 0:
            if st["skip"] > 0:
                st["skip"] -= 1
            elif not st["pending"] and not st["ms"]:
                # (a host that runs many frames ahead of the GPU finds the previous pair of measurements still
                # unfinished here: it waits for them instead of starting over)
                st["pending"]["other"] = []
                st["await_used"] = True
                return (not st["regen"]), True, st["pending"]["other"]
        if cycle > 1 and phase == 1 and st.get("await_used"):
            st["await_used"] = False
            st["pending"]["used"] = []
            return st["regen"], False, st["pending"]["used"]
        return st["regen"], False, None

    def _tile_schedule(self, rp: Tensor, rows, steps: int, regen: bool = False, record_now: bool = False):
        """(tile_order, tile_cost, after) of the next inference frame.  Every `adaptive_order`-th frame records the
        per-tile step counts and, right behind the frame on the same stream, sorts the tiles by decreasing cost
        (rm_tile_order_from_cost); the frames in between are dealt in that order.  Any order renders the same
        image, so a stale one (the camera moved) only costs the gain."""
        n, h, w, _ = rp.shape
        r0, r1 = rows if rows is not None else (0, h)
        dev = rp.device
        per_ray = regen and self.order_per_ray
        key = (r0, r1, steps, self.tile8x8, regen, per_ray, dev.index, torch.cuda.current_stream(dev).cuda_stream)
        st = self._order_state.get(key)
        if st is None:
            st = self._order_state[key] = self._new_order_state(rp, rows, regen)
        if not st["T"]:
            return None, None, None
        st["frame"] += 1
        order = st["order"] if st["valid"] else None
        if (st["frame"] - 1) % self.adaptive_order and not record_now:
            return order, None, None
        # the frame that records the cost reads the OLD order while the sort kernel that follows it on the same
        # stream writes the new one in place: stream order makes that safe
        return order, st["cost"], lambda: self._renew_order(st, steps, dev)

    def _new_order_state(self, rp: Tensor, rows, regen: bool):
        """Buffers of one dealing order ({"T": 0} where none is kept)."""
        n, h, w, _ = rp.shape
        r0, r1 = rows if rows is not None else (0, h)
        dev = rp.device
        per_ray = regen and self.order_per_ray
        T = int(ops._lib.rm_wave_tiles(n, r1 - r0, w, ops.default_flags(self.early_out, self.tile8x8, self.dynamic_tiles)))
        # Where it pays (measured, profiles/ab_probe.py): the permutation costs every tile one more dependent L2
        # round trip (order[position] before the ray loads: +18 us on the 1080p scene-2 frame, which gains
        # nothing from it), and it shortens the tail of launches whose longest tiles are a large part of a wave's
        # whole share: few tiles per wave (512^2 closed scene 1: 173 -> 118 us) or tiles whose per-step cost
        # varies with the scene part they hit (32-primitive 8K band: 13.6 -> 11.1 ms).
        # With hundreds of tiles per wave (the whole 8K frame in one launch) the tail is negligible again and the
        # lookups only cost: 84.7 -> 89.9 ms.
        # Ray regeneration always wants it: once the queues are dry a pool's idle lanes stay idle, so the tiles
        # with the longest rays have to go first.
        worth = regen or T <= 16384 or (compiled_for(self.scene).n_instr >= 64 and T <= 131072)
        if T < 4096 or not worth:
            return {"T": 0}
        # RM_FLAG_REGEN records one cost per ray slot (include/rm_abi.h) and is dealt by tile scores made of
        # them, or -- order_per_ray -- by the ray costs themselves
        n_cost = T * 64 if regen else T
        n_order = T * 64 if per_ray else T
        return {"T": T, "n_order": n_order, "cost": torch.empty(n_cost, dtype=torch.int32, device=dev),
                "order": torch.empty(n_order, dtype=torch.int32, device=dev), "frame": 0, "valid": False,
                "score": torch.empty(2 * T, dtype=torch.int32, device=dev) if (regen and not per_ray) else None,
                "grid": ((w + 7) // 8, (r1 - r0 + 7) // 8),
                "scratch": torch.empty(_abi.ORDER_SCRATCH_INTS, dtype=torch.int32, device=dev)}

    @staticmethod
    def _renew_order(st, steps: int, dev):
        """st["cost"] (written by the frame just launched on this stream) -> st["order"], on the same stream."""
        with torch.cuda.device(dev):
            stream = _abi.current_stream(dev)
            src, top = st["cost"], steps
            if st["score"] is not None:
                raw = st["score"][st["T"]:]
                _abi.check(ops._lib.rm_tile_score_from_ray_cost(_abi.ptr(st["cost"]), st["T"], st["grid"][0], st["grid"][1],
                                                                RenderLoop.ORDER_REACH_TILES, steps, _abi.ptr(raw),
                                                                _abi.ptr(st["score"]), stream), "rm_tile_score_from_ray_cost")
                src, top = st["score"], 31
            _abi.check(ops._lib.rm_tile_order_from_cost(_abi.ptr(src), st["n_order"], top, _abi.ptr(st["order"]),
                                                        _abi.ptr(st["scratch"]), stream), "rm_tile_order_from_cost")
        st["valid"] = True

    def capture(self, mode: int = 0, degree: int = 1, marching_steps: int = 32, rows=None, display: bool = False) -> CapturedFrame:
        """HIP-graph replay of one inference frame (see CapturedFrame); ``display=True``: of the [H,W,4] display tensor."""
        return CapturedFrame(self, mode, degree, marching_steps, rows, display)

    def training_step(self, loss_fn, mode: int = 0, degree: int = 1, marching_steps: int = 32, optimizer=None,
                      pose_requires_grad: bool = False):
        """forward -> ``loss_fn(image)`` -> backward (-> ``optimizer.step()``) captured into ONE HIP graph on first use
        and replayed from then on (graphs.CapturedTrainingStep): ``step = loop.training_step(loss_fn, ...)``, then
        ``loss = step(orientations, translations)`` per iteration.  Takes the training loop off the host: an eager step
        of the config-4 shape is ~60 launches and 40 AccumulateGrad nodes for 0.40 ms of GPU work."""
        from .graphs import CapturedTrainingStep
        return CapturedTrainingStep(self, loss_fn, mode, degree, marching_steps, optimizer, pose_requires_grad)

    def display_frame(self, orientations: Tensor, translations: Tensor, mode: int = 0, degree: int = 1,
                      marching_steps: int = 32) -> Tensor:
        """What main.py:78-84 hands to ``Window.draw`` -- ``F.pad(images.mean(dim=0).float(), [0, 1], value=1.0)``:
        contiguous [H, W, 4] float32, alpha 1 (torchwindow/window.py:146-174) -- written by the frame kernel itself
        instead of by three tensor passes over the image (mean, cast, pad).  One camera, inference only; bit-identical
        with the three passes (tests/test_gpu_round3.py).  Batches of cameras go through ``forward`` + headless.to_rgba."""
        with torch.no_grad():
            return self.forward(orientations, translations, mode, degree, marching_steps, _image_dtype="rgba")

    @torch.compiler.disable      # main.py:44 wraps the loop in torch.compile: Dynamo steps over the ctypes launches (eager bits, no Inductor kernel)
    def forward(self, orientations: Tensor, translations: Tensor, mode: int = 0, degree: int = 1,
                marching_steps: int = 32, rows=None, allreduce_minmax=None, tile_order=None, tile_cost=None,
                _image_dtype=None):
        """-> image [N, H, W, 3] in the module's dtype (modes 6, 7: promoted with the colormap's, float64 for
        the reference's data file).  ``rows=(r0, r1)`` renders only that pixel-row band ([N, r1-r0, W, 3]);
        ``allreduce_minmax`` is the hook row-tiled multi-GPU rendering uses for the global min/max of modes
        1/2/5 (see ray_marching_amd/distributed.py); ``tile_order`` / ``tile_cost``: rm_render_forward's
        scheduling hint and per-tile cost output (include/rm_abi.h)."""
        mode = mode % 8
        r0 = self.camera.rows[0]
        if rows is not None and r0:          # a band loop: frame rows -> rows of its own buffers
            if rows[0] < r0 or rows[1] > self.camera.rows[1]:
                raise ValueError(f"rows {tuple(rows)} outside this loop's band {self.camera.rows}")
            rows = (rows[0] - r0, rows[1] - r0)
        cs = compiled_for(self.scene)
        training = torch.is_grad_enabled() and (orientations.requires_grad or translations.requires_grad
                                                or any(p.requires_grad for p in cs.leaves))
        rp, rd = self._io_buffers(training)
        ops._require_device(rp, "camera buffers")          # no CPU path: fail here, with the reason, not in a stream query
        ops._require_device(orientations, "orientations")
        cmap = self._cmap(rp.device) if mode in (6, 7) else None
        # The kernels gather the parameter block from the nn.Parameter storages themselves (nothing to pack,
        # nothing to go stale); a training frame hands the Parameters to the autograd Function as its leaves.
        # Only parameters that cannot be read in place (other device / dtype) are packed with torch.cat.
        steps = int(marching_steps)
        regen, record_now, events = False, False, None
        if self.regen and tile_order is None and tile_cost is None and ops.regen_applies(
                ops.default_flags(self.early_out, self.tile8x8, self.dynamic_tiles, True), steps, training):
            if self.regen == "auto" and (self.precision != "exact" or torch.cuda.is_current_stream_capturing()):
                # the "fast" build contracts FMAs per kernel: switching kernels between frames could flicker in the last
                # bit; and a frame being captured into a HIP graph cannot carry timing events or a choice made later
                pass
            elif self.regen == "auto":
                band = rows if rows is not None else (0, rp.shape[1])
                if rp.shape[0] * (band[1] - band[0]) * rp.shape[2] >= self.REGEN_AUTO_MIN_RAYS:
                    regen, record_now, events = self._choose_kernel(rp, rows, steps)
            else:
                regen = True
        flags = ops.default_flags(self.early_out, self.tile8x8, self.dynamic_tiles, regen, self.order_per_ray)
        after = None
        if (self.adaptive_order > 0 and tile_order is None and tile_cost is None and self.dynamic_tiles
                and self.early_out):
            tile_order, tile_cost, after = self._tile_schedule(rp, rows, steps, regen, record_now)
        params, leaves = None, ()
        if training or _AB_PACK:
            if _AB_PACK:
                params = cs.pack_params(rp.device)
            else:       # parameters that cannot be read in place are packed (without grad) inside scene_struct;
                leaves = cs.leaves      # their gradients still go to the leaves
        image = ops.render_frame(params, orientations, translations, cs, rp, rd,
                                 self.normals.tetra(), cmap, mode, int(degree), int(marching_steps), rows,
                                 flags, allreduce_minmax, self.precision, _image_dtype, tile_order, tile_cost, leaves, events)
        if after is not None:
            after()
        if _image_dtype == "rgba":
            return image
        out_dtype = self.camera.ray_positions.dtype
        if mode in (6, 7):
            out_dtype = torch.promote_types(out_dtype, self.shader.cyclic_cmap.dtype)
        return image if image.dtype == out_dtype else image.to(out_dtype)
