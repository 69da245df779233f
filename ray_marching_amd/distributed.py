"""Row-tile sharding of a frame over the GPUs of one node.

Rays are independent, so a frame is split into contiguous pixel-row bands, one per rank (one
process per GPU, ``torch.distributed`` backend "nccl" = RCCL over xGMI).  The scene block is
replicated (a few hundred bytes).  The only exchanges are:

* a 2-float all-reduce(min/max) between the two passes of the globally normalised shaders
  (modes 1, 2, 5; reference rendering/shader.py:35-36, 52-53, 84);
* the final gather of the image tiles (to one rank, or all-gather);
* a P-float all-reduce(sum) of the scene-parameter gradients in training.

Per-pixel shaders (modes 0, 3, 4, 6, 7) involve no exchange and the reassembled frame is
bit-identical to a single-GPU render (tests/test_gpu_parity.py::test_full_size_properties).

``render_fn`` is the band renderer: by default the HIP ``RenderLoop``.  Tests inject a CPU
function so the sharding / collective logic is exercised with gloo on machines without GPUs.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


# Shaders whose three channels are one value (the reference computes [N,H,W,1] and `.expand`s it,
# rendering/shader.py:20, 38, 55, 66, 89): lambertian, distance, proximity, vignette, laplacian.  Their tiles
# cross xGMI as one channel -- a third of the bytes, losslessly -- and are expanded on the receiving side.
GREY_MODES = (0, 1, 2, 3, 5)


def tile_payload(tile: torch.Tensor, mode: int) -> torch.Tensor:
    """What a rank sends for its tile [N, rows, W, 3]: channel 0 alone for the grey shaders."""
    return tile[..., :1].contiguous() if mode % 8 in GREY_MODES else tile.contiguous()


def expand_payload(frame: torch.Tensor) -> torch.Tensor:
    """[N,H,W,1] -> the [N,H,W,3] view the reference's Shader.forward returns; 3-channel frames pass."""
    return frame.expand(-1, -1, -1, 3) if frame.shape[-1] == 1 else frame


def row_band(height: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous band [r0, r1) of ``rank``: ceil(height / world) rows each, the last ones shorter
    (possibly empty) so that every gathered tile can be padded to one common size."""
    per = -(-height // world)
    r0 = min(rank * per, height)
    return r0, min(r0 + per, height)


class RowTileRenderer:
    """``exchange``: how the tiles reach the gathering rank.
      "p2p"    (default) one point-to-point transfer per peer, all posted together
               (``dist.batch_isend_irecv``): on xGMI every peer has its own link to the root, so the 7 tiles
               of an 8-GPU node arrive over 7 links at once instead of one ring;
      "gather" ``dist.gather`` / ``dist.all_gather`` -- the collective the p2p path is checked against."""

    def __init__(self, loop=None, group=None, render_fn: Optional[Callable] = None, height: Optional[int] = None,
                 exchange: str = "p2p", width: Optional[int] = None, num_cameras: Optional[int] = None, device=None):
        if loop is None and render_fn is None:
            raise ValueError("need a RenderLoop or a render_fn")
        if exchange not in ("p2p", "gather"):
            raise ValueError("exchange must be 'p2p' or 'gather'")
        self.loop = loop
        self.group = group
        self.exchange = exchange
        self.render_fn = render_fn if render_fn is not None else self._hip_band
        self.height = height if height is not None else loop.px_height
        # shape / placement of an EMPTY tile (a surplus rank has nothing to render but still joins every exchange)
        cam = getattr(loop, "camera", None)
        self.width = width if width is not None else (loop.px_width if loop is not None else None)
        self.num_cameras = num_cameras if num_cameras is not None else (cam.ray_positions.shape[0] if cam is not None else 1)
        self.device = torch.device(device) if device is not None else (cam.ray_positions.device if cam is not None else torch.device("cpu"))
        self.dtype = cam.ray_positions.dtype if cam is not None else torch.float32
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    # -- default band renderer: the fused HIP frame kernel on this rank's rows -------------
    def _hip_band(self, orientations, translations, mode, degree, steps, rows, allreduce_minmax):
        return self.loop(orientations, translations, mode, degree, steps, rows=rows,
                         allreduce_minmax=allreduce_minmax)

    def _host_staged(self, t: torch.Tensor) -> bool:
        """gloo moves host memory: device tensors are staged through the host for it (rehearsals of the N > 1
        path on a box without RCCL peers; with backend "nccl" = RCCL the device buffers go over xGMI directly)."""
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    # -- global min/max of the normalised shaders -------------------------------------------
    def _allreduce_minmax(self, lohi: torch.Tensor):
        """lohi = [min, max] of this rank's band, reduced in place over the group: ONE all-reduce(MIN) of
        [min, -max] (negation is exact; -(-x) restores every bit, inf and NaN included)."""
        if self.world == 1:
            return
        staged = self._host_staged(lohi)
        pair = (lohi.cpu() if staged else lohi).clone()
        pair[1].neg_()
        dist.all_reduce(pair, op=dist.ReduceOp.MIN, group=self.group)
        pair[1].neg_()
        lohi.copy_(pair)

    def band(self):
        return row_band(self.height, self.rank, self.world)

    def render_band(self, orientations, translations, mode: int = 0, degree: int = 1, marching_steps: int = 32):
        """This rank's tile [N, rows, W, 3] (rows may be 0 for surplus ranks)."""
        r0, r1 = self.band()
        if r1 <= r0:
            return None
        return self.render_fn(orientations, translations, mode, degree, marching_steps, (r0, r1),
                              self._allreduce_minmax)

    def _empty_tile(self, like):
        if like is not None:
            return like.new_zeros((like.shape[0], 0, like.shape[2], 3))
        if self.width is None:
            raise ValueError("a rank without rows needs the frame width (RowTileRenderer(width=...)) or `like`")
        return torch.zeros((self.num_cameras, 0, self.width, 3), dtype=self.dtype, device=self.device)

    def _exchange(self, padded: torch.Tensor, dst: Optional[int]):
        """The tiles of all ranks, stacked along the row axis in rank order ([N, per * world, W, C]), on rank ``dst``
        (None elsewhere), or on every rank for dst=None.  With one camera every tile is received straight into its row
        slice of ONE preallocated frame (no concatenation pass: 0.4-0.8 GB of extra traffic at 8K); camera batches,
        whose row slices are not contiguous, arrive in separate tensors and are concatenated."""
        if self._host_staged(padded):
            device = padded.device
            frame = self._exchange(padded.cpu(), dst)
            return None if frame is None else frame.to(device)
        n, per, w, c = padded.shape
        receives = dst is None or self.rank == dst
        frame = parts = None
        if receives and n == 1:
            frame = padded.new_empty((1, per * self.world, w, c))
            parts = [frame[:, r * per:(r + 1) * per] for r in range(self.world)]     # contiguous views
        elif receives:
            parts = [torch.empty_like(padded) for _ in range(self.world)]
        if dst is None:
            dist.all_gather(parts, padded, group=self.group)
        elif self.exchange == "gather":
            dist.gather(padded, parts, dst=dst, group=self.group)
        else:
            # point to point: the root posts one receive per peer, every peer one send, all in one batch
            if self.rank == dst:
                parts[dst].copy_(padded)
                ops = [dist.P2POp(dist.irecv, parts[r], r, group=self.group) for r in range(self.world) if r != dst]
            else:
                ops = [dist.P2POp(dist.isend, padded, dst, group=self.group)]
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if not receives:
            return None
        return frame if frame is not None else torch.cat(parts, dim=1)

    def render(self, orientations, translations, mode: int = 0, degree: int = 1, marching_steps: int = 32,
               dst: Optional[int] = 0, like: Optional[torch.Tensor] = None):
        """Render this rank's band and gather the frame.  ``dst=None`` all-gathers (every rank gets
        the frame); otherwise only rank ``dst`` returns it and the others return None."""
        tile = self.render_band(orientations, translations, mode, degree, marching_steps)
        if self.world == 1:
            return tile
        per = -(-self.height // self.world)
        if tile is None:
            # surplus rank: nothing to render, but it takes part in every exchange (incl. the min/max all-reduce);
            # everything it needs is known BEFORE the first collective, so it cannot fail while the others wait
            tile = self._empty_tile(like)
            if mode % 8 in (1, 2, 5):
                lohi = torch.tensor([float("inf"), float("-inf")], device=tile.device)
                self._allreduce_minmax(lohi)
        tile = tile_payload(tile, mode)
        n, rows, w, c = tile.shape
        padded = tile if rows == per else torch.cat([tile, tile.new_zeros((n, per - rows, w, c))], dim=1)
        frame = self._exchange(padded.contiguous(), dst)
        if frame is None:
            return None
        return expand_payload(frame[:, : self.height])


def all_reduce_gradients(module: torch.nn.Module, group=None):
    """Sum the scene-parameter gradients over the ranks (each rank back-propagated the loss of
    its own band).  One flat P-float all-reduce."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    params = [p for p in module.parameters() if p.requires_grad]
    if not params:
        return
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    o = 0
    for p in params:
        n = p.numel()
        p.grad = flat[o:o + n].view_as(p).clone()
        o += n
