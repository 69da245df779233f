"""autograd.Function wrappers over the C ABI (librm_hip.so).

Each Function hands raw device pointers and the current HIP stream to one entry
point of include/rm_abi.h.  PyTorch only supplies device memory, the stream and
the autograd tape; all arithmetic happens in the HIP kernels.  Inputs must live
on a HIP device: there is deliberately no CPU path.
"""
from __future__ import annotations

import os
import warnings

import torch

from . import _abi
from .compiler import CompiledScene

_lib = _abi.lib


def _require_device(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"ray_marching_amd: {what} is on {t.device}; the HIP kernels need a ROCm device tensor "
            "(no CPU fallback exists by design)")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


_warned_f64: list = []
use_scene_cache = os.environ.get("RM_SCENE_CACHE", "1") != "0"      # A/B knob


def scene_cache(cs: CompiledScene, dev, stream):
    """The validated scene-block cache of (scene, device, stream) -- launches on one stream are ordered, so they may share
    one; created filled with 0xFFFFFFFF words (no parameter block compares equal to that).  None under graph capture
    when none exists yet (its initialisation would be replayed with the graph), and for parameter-free scenes."""
    if not use_scene_cache or cs.n_params <= 0:
        return None
    caches = cs.__dict__.setdefault("_block_caches", {})
    key = (dev.index, int(getattr(stream, "value", stream) or 0))
    buf = caches.get(key)
    if buf is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        buf = torch.full((cs.n_params + cs.n_derived,), -1, dtype=torch.int32, device=dev).view(torch.float32)
        caches[key] = buf
    return buf
bwd_clear_flags = _abi.FLAG_DYNAMIC_TILES if os.environ.get("RM_BWD_STATIC_TILES", "0") == "1" else 0      # A/B knob
use_forward_block = os.environ.get("RM_FORWARD_BLOCK", "1") != "0"      # A/B knob: backward kernels take the forward's scene block


def _io(t: torch.Tensor):
    """Contiguous tensor the kernels can read as is (fp32 or fp16; anything else is converted to fp32)
    and its RM_DTYPE_* code."""
    t = t.detach()
    if t.dtype not in (torch.float32, torch.float16):
        if t.dtype == torch.float64 and not _warned_f64:
            _warned_f64.append(True)
            warnings.warn("ray_marching_amd: float64 input -- the HIP kernels compute in float32 (the reference would "
                          "evaluate a .double() scene in float64); the result is cast back to float64", stacklevel=3)
        t = t.float()
    return t.contiguous(), _abi.dtype_code(t.dtype)


def _partials(cs: CompiledScene, s, device):
    """Workspace of the backward kernels' per-block partial sums for the scene struct ``s``."""
    n = cs.lib(True).rm_grad_partials_floats(s, 0)
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=device)


def _ck(cs, code, what):
    _abi.check(code, what, cs.lib())


def live_params(cs: CompiledScene, device, *inputs):
    """The packed parameter block (torch.cat: the autograd edge to every nn.Parameter) when a gradient can be
    asked for, else None: the kernels then gather the parameters from their storages (CompiledScene.param_table)."""
    if torch.is_grad_enabled() and (any(p.requires_grad for p in cs.leaves) or
                                    any(t is not None and t.requires_grad for t in inputs)):
        return cs.pack_params(device)
    return None


def default_flags(early_out: bool = True, tile8x8: bool = False, dynamic_tiles: bool = False, regen: bool = False,
                  order_per_ray: bool = False) -> int:
    return ((_abi.FLAG_EARLY_OUT if early_out else 0) | (_abi.FLAG_TILE8X8 if tile8x8 else 0) |
            (_abi.FLAG_DYNAMIC_TILES if dynamic_tiles else 0) | (_abi.FLAG_REGEN if regen else 0) |
            (_abi.FLAG_ORDER_PER_RAY if (regen and order_per_ray) else 0))


def regen_applies(flags: int, steps: int, record: bool) -> bool:
    """Whether a frame with these flags can take the ray-regeneration kernels (include/rm_abi.h, RM_FLAG_REGEN)."""
    need = _abi.FLAG_REGEN | _abi.FLAG_EARLY_OUT | _abi.FLAG_TILE8X8 | _abi.FLAG_DYNAMIC_TILES
    return (flags & need) == need and not record and steps > 0 and steps % 4 == 0


# --------------------------------------------------------------------------
# scene(query) -> distance
# --------------------------------------------------------------------------
class SDFEval(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, points, cs: CompiledScene):
        _require_device(points, "query_positions")
        dev = points.device
        pts, dt = _io(points)
        pts = pts.reshape(-1, 3)
        prm = None if params is None else _f32c(params)
        n = pts.shape[0]
        out = torch.empty(n, dtype=pts.dtype, device=dev)
        with torch.cuda.device(dev):
            s, keep = cs.scene_struct(prm, dev)
            _ck(cs, cs.lib().rm_sdf_forward(s, _abi.ptr(pts), _abi.ptr(out), n, dt, _abi.current_stream(dev)),
                       "rm_sdf_forward")
        ctx.cs = cs
        ctx.in_dtype = points.dtype
        if prm is not None:
            ctx.save_for_backward(prm, pts)
        return out.view(*points.shape[:-1], 1).to(points.dtype)

    @staticmethod
    def backward(ctx, grad_out):
        prm, pts = ctx.saved_tensors
        pts = pts.float()                      # the backward kernels are fp32 only
        cs, dev = ctx.cs, pts.device
        n = pts.shape[0]
        if n == 0:
            return torch.zeros_like(prm), torch.zeros(*grad_out.shape[:-1], 3, dtype=ctx.in_dtype, device=dev), None
        g = _f32c(grad_out).reshape(-1)
        gpts = torch.empty_like(pts) if ctx.needs_input_grad[1] else None
        gprm = torch.empty(max(cs.n_params, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            s, keep = cs.scene_struct(prm, dev)
            part = _partials(cs, s, dev)
            _ck(cs, cs.lib(True).rm_sdf_backward(s, _abi.ptr(pts), _abi.ptr(g), _abi.ptr(gpts), _abi.ptr(gprm),
                                            _abi.ptr(part), n, _abi.current_stream(dev)), "rm_sdf_backward")
        gp = gpts.view(*grad_out.shape[:-1], 3).to(ctx.in_dtype) if gpts is not None else None
        return gprm[: prm.numel()], gp, None


# --------------------------------------------------------------------------
# marcher
# --------------------------------------------------------------------------
class March(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, pos, dirs, cs: CompiledScene, steps: int, flags: int):
        _require_device(pos, "ray_positions")
        _require_device(dirs, "ray_directions")
        dev = pos.device
        in_dtype = pos.dtype
        shape = torch.broadcast_shapes(pos.shape, dirs.shape)
        need_grad = any(ctx.needs_input_grad[:3])
        if need_grad or pos.dtype != dirs.dtype:
            pos, dirs = pos.float(), dirs.float()          # training path: fp32 throughout
        p, dt = _io(pos.expand(shape))
        v, _ = _io(dirs.expand(shape))
        p, v = p.reshape(-1, 3), v.reshape(-1, 3)
        prm = None if params is None else _f32c(params)
        n = p.shape[0]
        out = torch.empty_like(p)
        traj = torch.empty((steps, n, 3), dtype=torch.float32, device=dev) if (need_grad and steps > 0) else None
        nexec = torch.empty(n, dtype=torch.int32, device=dev) if need_grad else None
        with torch.cuda.device(dev):
            s, keep = cs.scene_struct(prm, dev)
            _ck(cs, cs.lib().rm_march_forward(s, _abi.ptr(p), _abi.ptr(v), _abi.ptr(out), _abi.ptr(traj),
                                             _abi.ptr(nexec), n, steps, flags, dt, _abi.current_stream(dev)),
                       "rm_march_forward")
        ctx.cs, ctx.steps, ctx.shape, ctx.in_dtype = cs, steps, shape, in_dtype
        ctx.pos_shape, ctx.dirs_shape = pos.shape, dirs.shape
        if need_grad:
            ctx.save_for_backward(prm, v, traj, nexec)
        return out.view(shape).to(in_dtype)

    @staticmethod
    def backward(ctx, grad_out):
        prm, v, traj, nexec = ctx.saved_tensors
        cs, dev, steps = ctx.cs, v.device, ctx.steps
        n = v.shape[0]
        g = _f32c(grad_out.expand(ctx.shape)).reshape(-1, 3)
        if steps == 0 or n == 0:
            return torch.zeros_like(prm), grad_out, torch.zeros_like(grad_out), None, None, None
        gpos = torch.empty_like(v)
        gdirs = torch.empty_like(v) if ctx.needs_input_grad[2] else None
        gprm = torch.empty(max(cs.n_params, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            s, keep = cs.scene_struct(prm, dev)
            part = _partials(cs, s, dev)
            _ck(cs, cs.lib(True).rm_march_backward(s, _abi.ptr(v), _abi.ptr(traj), _abi.ptr(nexec), _abi.ptr(g),
                                              _abi.ptr(gpos), _abi.ptr(gdirs), _abi.ptr(gprm), _abi.ptr(part),
                                              n, steps, _abi.current_stream(dev)), "rm_march_backward")

        def fit(t, shape):
            t = t.view(ctx.shape).to(ctx.in_dtype)
            return t.sum_to_size(shape) if tuple(shape) != tuple(ctx.shape) else t

        return (gprm[: prm.numel()], fit(gpos, ctx.pos_shape),
                fit(gdirs, ctx.dirs_shape) if gdirs is not None else None, None, None, None)


# --------------------------------------------------------------------------
# normals / laplacian
# --------------------------------------------------------------------------
def make_tetra(offsets: torch.Tensor, inverse: torch.Tensor, eps: float) -> _abi.RmTetra:
    t = _abi.RmTetra()
    o = offsets.detach().float().cpu().reshape(-1).tolist()
    m = inverse.detach().float().cpu().reshape(-1).tolist()
    for i in range(12):
        t.offsets[i] = o[i]
    for i in range(9):
        t.inverse[i] = m[i]
    t.lap_scale = float(torch.tensor(6 / eps ** 2, dtype=torch.float32))
    return t


class Normals(torch.autograd.Function):
    @staticmethod
    def forward(ctx, params, coords, cs: CompiledScene, tetra):
        _require_device(coords, "surface_coords")
        dev = coords.device
        pts, dt = _io(coords)
        pts = pts.reshape(-1, 3)
        prm = None if params is None else _f32c(params)
        n = pts.shape[0]
        nrm = torch.empty_like(pts)
        lap = torch.empty(n, dtype=pts.dtype, device=dev)
        with torch.cuda.device(dev):
            s, keep = cs.scene_struct(prm, dev)
            _ck(cs, cs.lib().rm_normals_forward(s, tetra, _abi.ptr(pts), _abi.ptr(nrm), _abi.ptr(lap), n, dt,
                                               _abi.current_stream(dev)), "rm_normals_forward")
        ctx.cs, ctx.tetra, ctx.in_dtype = cs, tetra, coords.dtype
        if prm is not None:
            ctx.save_for_backward(prm, pts)
        return nrm.view(coords.shape).to(coords.dtype), lap.view(*coords.shape[:-1], 1).to(coords.dtype)

    @staticmethod
    def backward(ctx, grad_n, grad_lap):
        prm, pts = ctx.saved_tensors
        pts = pts.float()
        cs, dev = ctx.cs, pts.device
        n = pts.shape[0]
        gn = _f32c(grad_n).reshape(-1, 3) if grad_n is not None else None
        gl = _f32c(grad_lap).reshape(-1) if grad_lap is not None else None
        gpts = torch.empty_like(pts) if ctx.needs_input_grad[1] else None
        gprm = torch.empty(max(cs.n_params, 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            s, keep = cs.scene_struct(prm, dev)
            part = _partials(cs, s, dev)
            _ck(cs, cs.lib(True).rm_normals_backward(s, ctx.tetra, _abi.ptr(pts), _abi.ptr(gn), _abi.ptr(gl),
                                                _abi.ptr(gpts), _abi.ptr(gprm), _abi.ptr(part), n,
                                                _abi.current_stream(dev)), "rm_normals_backward")
        shape = grad_n.shape if grad_n is not None else (*grad_lap.shape[:-1], 3)
        gp = gpts.view(shape).to(ctx.in_dtype) if gpts is not None else None
        return gprm[: prm.numel()], gp, None, None


# --------------------------------------------------------------------------
# camera
# --------------------------------------------------------------------------
def camera_struct(ray_positions, ray_directions):
    """RmCamera over the two buffers (same dtype, fp32 or fp16, contiguous)."""
    n, h, w, _ = ray_positions.shape
    if ray_positions.dtype != ray_directions.dtype:
        raise ValueError("ray_positions and ray_directions must have one dtype")
    return _abi.RmCamera(ray_positions=ray_positions.data_ptr(), ray_directions=ray_directions.data_ptr(),
                         num_cameras=n, height=h, width=w, dtype=_abi.dtype_code(ray_positions.dtype))


def _camera_backward(rp, rd, q, gpos, gdirs, rows, need_q=True, need_t=True):
    """grad_orientation [N,4], grad_translation [N,3] from per-ray gradients (rm_camera_backward)."""
    dev = rp.device
    n = rp.shape[0]
    gq = torch.empty((n, 4), dtype=torch.float32, device=dev) if need_q else None
    gt = torch.empty((n, 3), dtype=torch.float32, device=dev) if need_t else None
    part = torch.empty(n * _abi.CAMERA_BWD_BLOCKS * 7, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        cam = camera_struct(rp, rd)
        _abi.check(_lib.rm_camera_backward(cam, _abi.ptr(q), _abi.ptr(gpos), _abi.ptr(gdirs), _abi.ptr(gq), _abi.ptr(gt),
                                           _abi.ptr(part), rows[0], rows[1], _abi.current_stream(dev)),
                   "rm_camera_backward")
    return gq, gt


class Camera(torch.autograd.Function):
    """PinholeCamera.forward: world-frame ray origins / directions + the [N,3,3] rotation matrices."""

    @staticmethod
    def forward(ctx, orientation, translation, ray_positions, ray_directions):
        _require_device(ray_positions, "camera buffers")
        _require_device(orientation, "orientation")
        dev = ray_positions.device
        rp, _ = _io(ray_positions)
        rd = ray_directions.detach().to(rp.dtype).contiguous()
        q, t = orientation.detach().to(rp.dtype).contiguous(), translation.detach().to(rp.dtype).contiguous()
        n = rp.shape[0]
        if q.shape != (n, 4) or t.shape != (n, 3):
            raise ValueError(f"camera pose shapes {tuple(q.shape)}, {tuple(t.shape)} do not match num_cameras={n}")
        pos, dirs = torch.empty_like(rp), torch.empty_like(rd)
        frames = torch.empty((n, 3, 3), dtype=rp.dtype, device=dev)
        with torch.cuda.device(dev):
            cam = camera_struct(rp, rd)
            _abi.check(_lib.rm_camera_forward(cam, _abi.ptr(q), _abi.ptr(t), _abi.ptr(pos), _abi.ptr(dirs),
                                              _abi.ptr(frames), _abi.current_stream(dev)), "rm_camera_forward")
        ctx.save_for_backward(rp, rd, q)
        ctx.mark_non_differentiable(frames)
        dt = ray_positions.dtype
        ctx.dt = orientation.dtype
        return pos.to(dt), frames.to(dt), dirs.to(dt)

    @staticmethod
    def backward(ctx, gpos, gframes, gdirs):
        rp, rd, q = (x.float() for x in ctx.saved_tensors)
        gp = _f32c(gpos) if gpos is not None else None
        gd = _f32c(gdirs) if gdirs is not None else None
        if gp is None and gd is None:
            return None, None, None, None
        gq, gt = _camera_backward(rp, rd, q, gp, gd, (0, rp.shape[1]), ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        if gt is not None and gp is None:
            gt = torch.zeros_like(gt)
        return (gq.to(ctx.dt) if gq is not None else None, gt.to(ctx.dt) if gt is not None else None, None, None)


def camera_forward(ray_positions, ray_directions, orientation, translation):
    return Camera.apply(orientation, translation, ray_positions, ray_directions)


class _Workspaces:
    """Launch workspaces (RM_WORK_WORDS: min/max words, tile-queue counters) prepared a batch at a time by ONE launch
    per device and stream, instead of one 5-us launch in front of every frame or backward kernel.  A batch is a fresh
    allocation (callers may keep a workspace: measurement hooks, the min/max all-reduce), handed out in stream order
    on the stream it was prepared on.  Under HIP-graph capture every frame prepares its own, so a replay is
    self-contained."""
    BATCH = 16

    def __init__(self):
        self._pools = {}

    def take(self, dev, stream):
        if torch.cuda.is_current_stream_capturing():
            ws = torch.empty(_abi.WORK_WORDS, dtype=torch.int32, device=dev)
            _abi.check(_lib.rm_minmax_init(_abi.ptr(ws), stream), "rm_minmax_init")
            return ws
        key = (dev.index, stream.value)
        pool = self._pools.get(key)
        if pool is None and len(self._pools) >= 64:      # streams come and go: forget the batches of old ones
            self._pools.clear()
        if pool is None or pool[1] == self.BATCH:
            buf = torch.empty(self.BATCH * _abi.WORK_WORDS, dtype=torch.int32, device=dev)
            _abi.check(_lib.rm_minmax_init_many(_abi.ptr(buf), self.BATCH, stream), "rm_minmax_init_many")
            pool = self._pools[key] = [buf, 0]
        i = pool[1]
        pool[1] += 1
        return pool[0][i * _abi.WORK_WORDS:(i + 1) * _abi.WORK_WORDS]


workspaces = _Workspaces()


# --------------------------------------------------------------------------
# fused frame
# --------------------------------------------------------------------------
_GLOBAL_MODES = (1, 2, 5)
_FUSED_VJP_MODES = {0, 1, 2, 3, 4, 5, 6, 7}   # shader modes rm_render_backward differentiates through


def minmax_normalisation_vjp(grad_image: torch.Tensor, logd: torch.Tensor, lo: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    """VJP of the distance / proximity shaders' normalisation (shader.py:33-38, 51-55:
    log_dists.sub(min).div(max.sub(min)).pow(1 / 2.33), expanded to three channels) in the order autograd walks it;
    returns dL/d(log_dists) per ray.  The pixel of the minimum has x = 0 under the power, whose slope there is
    infinite: inf - inf reaches the ray of the minimum and inf * 0 the ray of the maximum, in the reference as here,
    and every parameter component those two rays reach gets a NaN gradient; the other rays' upstreams are finite."""
    gamma = 1.0 / 2.33
    gy = grad_image.sum(-1)
    r = hi - lo
    a = logd - lo
    gx = gy * (gamma * (a / r).pow(gamma - 1.0))
    ga = gx / r
    gr = (-gx * a / (r * r)).sum()
    gm = -ga.sum() - gr                   # both log_dists.min() nodes
    top, bottom = logd == hi, logd == lo
    zero = torch.zeros_like(ga)
    return ga + torch.where(top, gr / top.sum(), zero) + torch.where(bottom, gm / bottom.sum(), zero)


def laplacian_normalisation_vjp(grad_image: torch.Tensor, lap: torch.Tensor, hi: torch.Tensor) -> torch.Tensor:
    """VJP of LaplacianShader's normalisation (shader.py:81-89: lap / lap.abs().max() * -1 + 1, / 2, clamp(0, 1),
    pow(1 / 2.33), expanded to three channels) in the order autograd walks it; returns dL/d(surface_laplacian)
    per ray.  ``lap``: the un-normalised Laplacian [...], ``hi``: its largest magnitude over the frame (0-dim).
    Where the pixel of the largest |Laplacian| has a POSITIVE one the clamped value is 0 and x^(1/2.33) has an
    infinite slope there: that ray's upstream is inf - inf = NaN, in the reference as here, and every parameter
    component the ray reaches gets a NaN gradient."""
    gamma = 1.0 / 2.33
    gy = grad_image.sum(-1)
    u = ((lap / hi) * -1.0 + 1.0) / 2.0
    x = u.clamp(0.0, 1.0)
    gu = gy * (gamma * x.pow(gamma - 1.0)) * ((u >= 0.0) & (u <= 1.0))
    ga = gu * 0.5 * -1.0
    glap = ga / hi
    ghi = (ga * (-lap / (hi * hi))).sum()
    top = lap.abs() == hi               # max(): the gradient goes to the largest element(s), evenly (masked_fill on the CPU)
    return glap + torch.where(top, (ghi / top.sum()) * lap.sign(), torch.zeros_like(glap))

_N_FIXED_ARGS = 18                   # Render.forward arguments in front of *leaves


# measurement hook (bench.py): when set to a list, Render.run appends a (start, end) pair of timing events
# recorded on the launch stream immediately around the k_render_fwd launch
kernel_event_sink = None
# forward: park never-settling rays for a dense second kernel.  Needs libraries built with -DRM_PARKING
# (RM_HIPCC_EXTRA="-DRM_PARKING" RM_PARK=1); off by default: +12 % at the reference's pose, slower everywhere else
park_rays = os.environ.get("RM_PARK", "0") == "1"
bwd_hard_capacity = None      # None: default sizing of the deferred-ray list of rm_render_backward; 0 = off (A/B probes)
bwd_tile_cost_sink = None     # measurement hook (profiles/): int32 [wave tiles] tensor receiving rm_render_backward's tile_cost


class Render(torch.autograd.Function):
    """RenderLoop.forward as one kernel (+ the normalisation pass of modes 1, 2, 5).

    ``allreduce_minmax``: optional callable(lohi_tensor[2]) applied between the two passes
    (row-tiled multi-GPU rendering all-reduces the global min/max there).

    I/O types: the camera buffers' dtype (fp32 or fp16) is the kernel's load type, the pose is passed in the
    same type, ``image_dtype`` is its store type -- no cast passes.  With anything requiring grad the frame is
    fp32 (the backward kernels are)."""

    @staticmethod
    def forward(ctx, params, orientation, translation, cs: CompiledScene, ray_positions, ray_directions,
                tetra, cmap, mode: int, degree: int, steps: int, rows, flags: int, allreduce_minmax,
                precision: str = "exact", image_dtype=None, tile_order=None, tile_cost=None, *leaves):
        """``params``: the packed block (a differentiable torch.cat of the leaves), or None with the scene's
        nn.Parameters themselves as trailing ``leaves``: the kernels then gather the block from the parameter
        storages (CompiledScene.param_table) -- no pack kernel, and autograd sees ONE node in front of the
        leaves instead of a cat + 2 view nodes per leaf (0.1 ms of host time per config-4 step)."""
        return Render.run(ctx, ctx.needs_input_grad, params, orientation, translation, cs, ray_positions,
                          ray_directions, tetra, cmap, mode, degree, steps, rows, flags, allreduce_minmax, precision,
                          image_dtype, tile_order, tile_cost, leaves)

    @staticmethod
    def run(ctx, needs_input_grad, params, orientation, translation, cs: CompiledScene, ray_positions,
            ray_directions, tetra, cmap, mode: int, degree: int, steps: int, rows, flags: int, allreduce_minmax,
            precision: str = "exact", image_dtype=None, tile_order=None, tile_cost=None, leaves=(), event_sink=None):
        """Body of the forward pass.  ``ctx`` is None for inference frames, which skip the autograd
        machinery altogether (render_frame below)."""
        _require_device(ray_positions, "camera buffers")
        _require_device(orientation, "orientations")
        dev = ray_positions.device
        prm = None if params is None else _f32c(params)       # None: the kernel gathers from the parameter storages
        need_grad = any(needs_input_grad[:3]) or any(needs_input_grad[_N_FIXED_ARGS:])
        have_vjp = mode in _FUSED_VJP_MODES
        record = need_grad and have_vjp
        if record and (ray_positions.dtype != torch.float32 or ray_directions.dtype != torch.float32):
            ray_positions, ray_directions = ray_positions.float(), ray_directions.float()
        rp, _ = _io(ray_positions)
        rd = ray_directions.detach().to(rp.dtype).contiguous()
        q, t = orientation.detach().to(rp.dtype).contiguous(), translation.detach().to(rp.dtype).contiguous()
        n, h, w, _ = rp.shape
        if q.shape != (n, 4) or t.shape != (n, 3):
            raise ValueError(f"pose shapes {tuple(q.shape)}, {tuple(t.shape)} do not match num_cameras={n}")
        r0, r1 = rows if rows is not None else (0, h)
        nrows = r1 - r0
        R = n * nrows * w
        rgba = image_dtype == "rgba"        # the display contract [rows, W, 4] fp32 (include/rm_abi.h: RM_DTYPE_RGBA_F32)
        if rgba and (record or n != 1):
            raise ValueError("the RGBA display frame is an inference output of ONE camera")
        if record:
            image_dtype = torch.float32
        elif image_dtype is None:
            image_dtype = torch.promote_types(rp.dtype, cmap.dtype) if (mode in (6, 7) and cmap is not None) else rp.dtype
        image = torch.empty((nrows, w, 4), dtype=torch.float32, device=dev) if rgba \
            else torch.empty((n, nrows, w, 3), dtype=image_dtype, device=dev)
        image_code = _abi.DTYPE_RGBA_F32 if rgba else _abi.dtype_code(image_dtype)
        first_pass = None
        if mode in _GLOBAL_MODES:
            # (a training frame of the Laplacian shader keeps the un-normalised values for its backward)
            first_pass = image if (image_dtype == torch.float32 and not record) \
                else torch.empty((n, nrows, w, 3), dtype=torch.float32, device=dev)
        regen = regen_applies(flags, steps, record)
        if not regen:
            flags &= ~(_abi.FLAG_REGEN | _abi.FLAG_ORDER_PER_RAY)
        # training: saved for the reverse sweep; ray regeneration: where the march kernel leaves the final iterates
        p_final = torch.empty((n, nrows, w, 3), dtype=torch.float32, device=dev) if (record or regen) else None
        # the trajectory (layout private to rm_render_forward / rm_render_backward) and the un-normalised normals
        traj = torch.empty(int(_lib.rm_render_traj_floats(n, nrows, w, steps, flags)), dtype=torch.float32, device=dev) \
            if (record and steps > 0) else None
        nexec = torch.empty(R, dtype=torch.int32, device=dev) if record else None
        normal_u = torch.empty((n, nrows, w, 3), dtype=torch.float32, device=dev) if record else None
        # parking workspace (rays that never settle are finished by a dense second kernel): room for 2 R rays
        # over the 32 list segments; inference frames of >= 48 steps with the early-out only
        park_cap = 0
        if park_rays and not record and not regen and steps >= 48 and (flags & _abi.FLAG_EARLY_OUT) and R >= 4096:
            park_cap = 2 * R
        park = torch.empty(int(_lib.rm_park_floats(park_cap)), dtype=torch.float32, device=dev) if park_cap else None
        with torch.cuda.device(dev):
            stream = _abi.current_stream(dev)
            # a training frame leaves its finished scene block (parameters + derived constants) for its backward
            # kernels, which then skip the gather + derive_constants prologue (RmScene.block, include/rm_abi.h)
            # (the ray pools' second kernel takes the block of their first one the same way)
            block_out = torch.empty(max(cs.n_params + cs.n_derived, 1), dtype=torch.float32, device=dev) \
                if (record or regen) and use_forward_block else None
            # inference frames: the derived constants of the previous launch on this stream, reused by the kernels when the
            # parameters they gather are bit for bit the ones the cache was derived from (RmScene.block_cache)
            cache = None if record else scene_cache(cs, dev, stream)
            s, keep = cs.scene_struct(prm, dev, block_out=block_out, block_cache=cache)
            cam = camera_struct(rp, rd)
            minmax = workspaces.take(dev, stream)      # global min/max words + tile-queue counters
            sink = event_sink if event_sink is not None else kernel_event_sink
            if sink is not None:
                ev0 = torch.cuda.Event(enable_timing=True)
                ev0.record()
            lib = cs.lib(False, precision)
            _abi.check(lib.rm_render_forward(s, cam, tetra, _abi.ptr(q), _abi.ptr(t), _abi.ptr(image),
                                             image_code, _abi.ptr(first_pass),
                                             _abi.ptr(p_final), _abi.ptr(traj), _abi.ptr(nexec), _abi.ptr(normal_u), _abi.ptr(minmax),
                                             _abi.ptr(cmap), 0 if cmap is None else cmap.shape[0],
                                             0 if cmap is None else _abi.dtype_code(cmap.dtype),
                                             mode, degree, steps, r0, r1, flags, _abi.ptr(tile_order), _abi.ptr(tile_cost),
                                             _abi.ptr(park), park_cap, stream), "rm_render_forward", lib)
            if sink is not None:
                ev1 = torch.cuda.Event(enable_timing=True)
                ev1.record()
                sink.append((ev0, ev1))
                if event_sink is None:
                    globals()["fwd_last_work"] = minmax   # measurement runs (kernel_event_sink): the workspace words of this frame
            lohi = None
            if mode in _GLOBAL_MODES:
                if allreduce_minmax is not None or record:
                    lohi = torch.empty(2, dtype=torch.float32, device=dev)
                    _abi.check(_lib.rm_minmax_decode(_abi.ptr(minmax), _abi.ptr(lohi), stream), "rm_minmax_decode")
                if allreduce_minmax is not None:
                    allreduce_minmax(lohi)
                    _abi.check(_lib.rm_minmax_encode(_abi.ptr(lohi), _abi.ptr(minmax), stream), "rm_minmax_encode")
                _abi.check(_lib.rm_shade_finish(_abi.ptr(first_pass), _abi.ptr(image), image_code,
                                                n * nrows * w, _abi.ptr(minmax), mode, _abi.dtype_code(rp.dtype), stream), "rm_shade_finish")
        if ctx is not None:
            ctx.have_vjp = have_vjp
            ctx.mode = mode
        if record:
            # the leaves are saved too: autograd then refuses a backward after an in-place edit of a parameter
            # (the backward kernels read the live storages, which must still hold the forward's values)
            ctx.save_for_backward(prm, q, t, rp, rd, p_final, traj, nexec, normal_u, *leaves)
            # Laplacian shader: the un-normalised values and their largest magnitude, for the normalisation's VJP
            ctx.lap = (first_pass, lohi, allreduce_minmax is not None) if mode in _GLOBAL_MODES else None
            ctx.scene_keep = keep       # (program, packed block or None, pointer table or None): what backward reads through
            ctx.cs, ctx.tetra, ctx.steps, ctx.rows, ctx.flags = cs, tetra, steps, (r0, r1), flags
            ctx.precision, ctx.cmap, ctx.degree = precision, cmap, degree
        return image

    @staticmethod
    def backward(ctx, grad_image):
        if not ctx.have_vjp:
            # the frame itself rendered (like the reference, which renders every mode with grad enabled);
            # only differentiating through a shader without a fused VJP is refused
            raise NotImplementedError(f"fused backward exists for shader modes {sorted(_FUSED_VJP_MODES)}, not {ctx.mode}")
        prm, q, t, rp, rd, p_final, traj, nexec, normal_u, *leaves = ctx.saved_tensors
        cs, dev = ctx.cs, rp.device
        g = _f32c(grad_image)
        if ctx.mode in _GLOBAL_MODES:
            # the normalisation over the whole frame is differentiated here; the kernels take dL/d(un-normalised value)
            raw3, lohi, across_ranks = ctx.lap
            if across_ranks:
                raise NotImplementedError("gradient of a globally normalised shader through a minimum / maximum taken across ranks")
            # rm_shade_norm_backward: two launches (block sums + the elementwise combination) in place of the ~15 tensor
            # ops of minmax_normalisation_vjp / laplacian_normalisation_vjp below, which stay as its readable statement
            graw3 = torch.empty_like(raw3)
            part = torch.empty(_abi.NORM_BWD_BLOCKS * 4, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _abi.check(_lib.rm_shade_norm_backward(_abi.ptr(raw3), _abi.ptr(g), _abi.ptr(lohi), ctx.mode, _abi.ptr(graw3),
                                                       _abi.ptr(part), raw3.numel() // 3, _abi.current_stream(dev)),
                           "rm_shade_norm_backward")
            g = graw3
        gprm = torch.empty(max(cs.n_params, 1), dtype=torch.float32, device=dev)
        lib = cs.lib(True, ctx.precision)
        with torch.cuda.device(dev):
            # the forward's own view of the parameters (its packed copy or pointer table): no second look at 40 storages
            s, keep = cs.scene_struct(prm if prm is not None else ctx.scene_keep[1], dev, table=ctx.scene_keep[2],
                                      block=ctx.scene_keep[4] if use_forward_block else None)
            cam = camera_struct(rp, rd)
            part = _partials(cs, s, dev)
            stream = _abi.current_stream(dev)
            work = workspaces.take(dev, stream)
            need_pose = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
            gpos = torch.empty_like(p_final) if need_pose else None
            gdirs = torch.empty_like(p_final) if ctx.needs_input_grad[1] else None
            # modes 3, 6, 7 use the pose quaternion in the shader itself: per-ray dL/dq, summed per camera below
            gqdir = torch.empty(p_final.shape[:-1] + (4,), dtype=torch.float32, device=dev) \
                if (ctx.needs_input_grad[1] and ctx.mode in (3, 6, 7)) else None
            cmap = ctx.cmap if ctx.mode in (6, 7) else None
            # deferred-ray workspace: room for one ray in eight (config 4 defers 2 %; the rest is walked in place)
            hard_cap = 0 if (bwd_hard_capacity == 0 or ctx.steps == 0) else \
                (bwd_hard_capacity or min(1 << 21, max(4096, p_final.numel() // 3 // 8)))
            hard = torch.empty(int(_lib.rm_bwd_hard_floats(hard_cap, ctx.steps)), dtype=torch.float32, device=dev) \
                if hard_cap else None
            if bwd_tile_cost_sink is not None:             # measurement runs: word 32 of `work` = rays deferred
                globals()["bwd_last_work"], globals()["bwd_last_hard"] = work, (hard, hard_cap)
            _abi.check(lib.rm_render_backward(s, cam, ctx.tetra, _abi.ptr(q), _abi.ptr(t), _abi.ptr(traj),
                                               _abi.ptr(nexec), _abi.ptr(p_final), _abi.ptr(normal_u), _abi.ptr(g), _abi.ptr(gprm),
                                               _abi.ptr(part), _abi.ptr(work), _abi.ptr(gpos), _abi.ptr(gdirs),
                                               _abi.ptr(gqdir), _abi.ptr(cmap), 0 if cmap is None else cmap.shape[0],
                                               0 if cmap is None else _abi.dtype_code(cmap.dtype),
                                               ctx.mode, ctx.degree, ctx.steps, ctx.rows[0], ctx.rows[1],
                                               ctx.flags & ~bwd_clear_flags,
                                               _abi.ptr(bwd_tile_cost_sink), _abi.ptr(hard), hard_cap, stream),
                       "rm_render_backward", lib)
        gq = gt = None
        if need_pose:
            gq, gt = _camera_backward(rp, rd, q, gpos, gdirs, ctx.rows, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
            if gqdir is not None:
                n_cam, per_cam = gqdir.shape[0], gqdir[0].numel() // 4
                direct = torch.empty((n_cam, 4), dtype=torch.float32, device=dev)
                with torch.cuda.device(dev):
                    for c in range(n_cam):
                        _abi.check(_lib.rm_sum_rows(_abi.ptr(gqdir[c]), per_cam, 4, _abi.ptr(direct[c]), stream), "rm_sum_rows")
                gq = gq + direct
        gp_out = gprm[: prm.numel()] if (prm is not None and ctx.needs_input_grad[0]) else None
        leaf_grads = []
        if leaves:                              # named_parameters() order = block order: one split, then views
            pieces = torch.split(gprm[: cs.n_params], cs.leaf_sizes)
            need = ctx.needs_input_grad
            for k, (p, g) in enumerate(zip(leaves, pieces)):
                if not need[_N_FIXED_ARGS + k]:
                    leaf_grads.append(None)
                    continue
                if p.dim() != 1:
                    g = g.view(p.shape)
                leaf_grads.append(g if p.dtype == torch.float32 else g.to(p.dtype))
        return (gp_out, gq, gt) + (None,) * (_N_FIXED_ARGS - 3) + tuple(leaf_grads)


_NO_GRAD = (False,) * 3


def render_frame(params, orientation, translation, cs: CompiledScene, ray_positions, ray_directions, tetra, cmap,
                 mode: int, degree: int, steps: int, rows, flags: int, allreduce_minmax, precision: str = "exact",
                 image_dtype=None, tile_order=None, tile_cost=None, leaves=(), event_sink=None):
    """One frame.  Goes through autograd only when something can receive a gradient; an inference frame
    calls the launch code directly (autograd.Function.apply costs ~20 us per call even under no_grad).
    ``event_sink`` (inference frames): a list that receives a (start, end) pair of timing events recorded on the
    launch stream immediately around rm_render_forward."""
    if torch.is_grad_enabled() and ((params is not None and params.requires_grad) or orientation.requires_grad
                                    or translation.requires_grad or any(p.requires_grad for p in leaves)):
        return Render.apply(params, orientation, translation, cs, ray_positions, ray_directions, tetra, cmap, mode,
                            degree, steps, rows, flags, allreduce_minmax, precision, image_dtype, tile_order, tile_cost,
                            *leaves)
    return Render.run(None, _NO_GRAD, params, orientation, translation, cs, ray_positions, ray_directions, tetra,
                      cmap, mode, degree, steps, rows, flags, allreduce_minmax, precision, image_dtype, tile_order,
                      tile_cost, (), event_sink)
