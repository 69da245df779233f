"""Scene compiler: nn.Module SDF tree -> flat program + packed parameter block.

The reference evaluates a scene by recursing through ``nn.Module.forward``
calls (scene/primitives.py, scene/transformations.py), one ATen op at a time.
Here the tree is walked once on the host and lowered to the instruction list of
``include/rm_abi.h`` (opcode, parameter offset, aux0, aux1), which the HIP
kernels evaluate per ray.

Parameter block layout = ``module.named_parameters()`` order (the reference's
state_dict order), so gradient vectors map 1:1 back onto the ``nn.Parameter``s.
Derived constants (capsule AB, AB/|AB|^2) get slots after the raw block; the
kernels fill them in LDS.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _abi

import os

# backward-pass stack cost of each frame kind, in floats (forward uses fewer)
_STACK_UNION, _STACK_SMOOTH, _STACK_AFFINE = 2, 2, 6


# rough VALU cost of evaluating a subtree (instructions per ray), used only to decide whether a cull test
# (~9 instructions when it fails) is worth emitting in front of it
_LEAF_COST = {"sphere": 13, "box": 22, "plane": 1, "line": 30, "disk": 20, "torus": 24}
_CULL_MIN_CHILD_COST = 40
# smooth unions with at least this many children get the exact logsumexp culling (RM_OP_CULL_LSE; the wave-wide test
# costs ~45 instructions per evaluation whatever the number of children)
_CULL_LSE_MIN_CHILDREN = 8


def _cost(node) -> int:
    kind = getattr(node, "_rm_kind", None)
    if kind in _LEAF_COST:
        return _LEAF_COST[kind]
    if kind == "affine":
        return 25 + _cost(node.sdf)
    if kind in ("rounding", "onion"):
        return 1 + _cost(node.sdf)
    if kind == "union":
        return sum(_cost(c) + 1 for c in node.sdfs)
    if kind == "smooth_union":
        return sum(_cost(c) + 25 for c in node.sdfs) + 20
    return 0


def _boundable(node) -> bool:
    """Can the kernels derive a bounding sphere for this subtree (csrc/rm_device.h: subtree_bound)?
    Everything except an SDFPlane somewhere inside; whether the bound is finite is decided on the device
    from the live parameter values."""
    kind = getattr(node, "_rm_kind", None)
    if kind in ("sphere", "box", "line", "disk", "torus"):
        return True
    if kind in ("affine", "rounding", "onion"):
        return _boundable(node.sdf)
    if kind in ("union", "smooth_union"):
        return all(_boundable(c) for c in node.sdfs)
    return False


@dataclass
class CompiledScene:
    program: np.ndarray                 # int32 [n_instr, 4]
    leaves: list                        # nn.Parameter objects in block order
    leaf_names: list
    leaf_offsets: list
    n_params: int
    n_derived: int
    n_grad_derived: int                 # leading derived floats that carry gradients (capsule constants)
    stack_floats: int
    n_slots: int
    signature: tuple                    # topology key (ops + offsets), parameters excluded
    _device_programs: dict = field(default_factory=dict)
    _table: dict = field(default_factory=dict)
    _leaf_sizes: object = None
    _lib: object = None

    # the loaded libraries (ctypes) and device tensors are process state, not scene state: a pickled / deep-copied
    # CompiledScene carries the program only and re-resolves the rest on first use
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_device_programs"], state["_table"], state["_lib"] = {}, {}, None
        state.pop("_block_caches", None)          # (ops.scene_cache: per-stream device buffers)
        return state

    def lib(self, backward: bool = False, precision: str = "exact"):
        """Kernel library for this scene: the per-scene specialised build when one is available
        (ray_marching_amd/specialize.py), else the generic interpreter library.  Backward of
        scenes with many parameters always uses the generic library (accumulators in LDS).
        ``precision="fast"`` selects the opt-in fast-arithmetic builds."""
        from . import specialize
        generic = _abi.generic_lib(precision)
        if backward and not specialize.static_backward(self):
            return generic
        if precision != "exact":
            return specialize.load(self, precision) or generic
        if self._lib is None:
            self._lib = specialize.load(self) or _abi.lib
        if self._lib is _abi.lib and specialize.note_interpreted_launch(self):
            self._lib = specialize.load(self) or _abi.lib      # a background build has finished
        return self._lib

    @property
    def specialised(self) -> bool:
        return self.lib() is not _abi.lib

    @property
    def n_instr(self):
        return int(self.program.shape[0])

    def device_program(self, device):
        key = str(device)
        t = self._device_programs.get(key)
        if t is None:
            t = torch.from_numpy(self.program.reshape(-1).copy()).to(device)
            self._device_programs[key] = t
        return t

    def pack_params(self, device):
        """fp32 parameter block on ``device`` as ONE tensor (torch.cat of the leaves: differentiable, so
        gradients flow back to every nn.Parameter).  Always rebuilt from the live values -- no cache that an
        in-place ``.data`` edit could leave stale.  Inference frames do not need it at all: see param_table."""
        leaves = self.leaves
        if not leaves:
            return torch.zeros(1, dtype=torch.float32, device=device)
        return torch.cat([p.reshape(-1).to(device=device, dtype=torch.float32) for p in leaves])

    @property
    def leaf_sizes(self):
        if self._leaf_sizes is None:
            self._leaf_sizes = [p.numel() for p in self.leaves]
        return self._leaf_sizes

    def param_table(self, device):
        """Device table of RmParamRef {pointer, element, dtype} for every float of the block, so the kernels
        gather the parameters from the nn.Parameter storages themselves (no packing pass; in-place edits,
        optimiser steps and ``.data`` writes are simply what the next launch reads).  The table only depends
        on where the leaves live -- (data_ptr, dtype), all visible on the host -- and is rebuilt when that
        changes.  None when a leaf cannot be read in place (not on ``device``, not fp32/fp16, not contiguous);
        the caller then packs."""
        leaves = self.leaves
        if not leaves:
            return None
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        key = (str(dev),) + tuple((p.data_ptr(), p.dtype) for p in leaves)
        hit = self._table.get("key")
        if hit == key:
            return self._table["value"]
        rows = np.zeros((self.n_params, 2), dtype=np.int64)        # {base pointer, elem | dtype << 32}
        at = 0
        for p in leaves:
            if p.device != dev or p.dtype not in (torch.float32, torch.float16) or not p.is_contiguous():
                self._table = {"key": key, "value": None}
                return None
            n = p.numel()
            rows[at:at + n, 0] = p.data_ptr()
            rows[at:at + n, 1] = np.arange(n, dtype=np.int64) | (np.int64(_abi.dtype_code(p.dtype)) << 32)
            at += n
        table = torch.from_numpy(rows.reshape(-1)).to(dev)
        self._table = {"key": key, "value": table}
        return table

    def scene_struct(self, params, device, table=None, block=None, block_out=None, block_cache=None):
        """RmScene for a launch; keeps the referenced tensors alive via the return tuple.  ``params``: the packed
        fp32 block, or None to gather from ``table`` (default: param_table(device)); packs when neither works.
        ``block``: the finished scene block (n_params + n_derived floats) a previous launch left in its ``block_out``:
        the kernels then skip gathering the parameters and deriving the constants.  ``block_cache``: a block the kernels
        reuse after comparing the parameters it was derived from with the live ones (ops.scene_caches)."""
        prog = self.device_program(device)
        if params is None and table is None:
            table = self.param_table(device)
            if table is None:
                with torch.no_grad():
                    params = self.pack_params(device)
        s = _abi.RmScene(program=prog.data_ptr(), params=None if params is None else params.data_ptr(),
                         param_refs=None if (params is not None or table is None) else table.data_ptr(),
                         n_instr=self.n_instr, n_params=self.n_params, n_derived=self.n_derived,
                         stack_floats=self.stack_floats, n_slots=self.n_slots, n_grad_derived=self.n_grad_derived,
                         block=None if block is None else block.data_ptr(),
                         block_out=None if block_out is None else block_out.data_ptr(),
                         block_cache=None if block_cache is None else block_cache.data_ptr())
        return s, (prog, params, table, block, block_out, block_cache)


class _Emitter:
    def __init__(self, offsets):
        self.offsets = offsets        # id(param) -> float offset
        self.code = []
        self.n_slots = 0
        self.n_derived = 0
        self.n_grad_derived = 0        # capsule constants: reserved at the front of the derived block (they carry gradients)
        self.next_line = 0
        self.depth = 0
        self.max_depth = 0
        self.cull = os.environ.get("RM_CULL", "1") != "0"    # RM_CULL=0: no CULL_MIN instructions (A/B tests)
        # RM_CULL_MIN_COST=0: a cull test in front of every boundable child, however cheap (stress tests)
        self.cull_min_cost = int(os.environ.get("RM_CULL_MIN_COST", _CULL_MIN_CHILD_COST))
        self.cull_reorder = os.environ.get("RM_CULL_REORDER", "1") != "0"
        # RM_CULL_LSE=1: exact culling inside smooth unions (opt-in: it pays when the children are spread over much more
        # than 104 / blend_k; in the config-5 scene 2.8 % of the child evaluations qualify, profiles/r03_lse_cull_rate.txt,
        # and the wave-wide test costs more than they save); RM_CULL_LSE_MIN: children from which on it is emitted
        self.cull_lse = os.environ.get("RM_CULL_LSE", "0") == "1"
        self.cull_lse_min = int(os.environ.get("RM_CULL_LSE_MIN", _CULL_LSE_MIN_CHILDREN))
        # RM_CULL_UNION_TABLE=0: a smooth union that is a cullable child of a min-union is tested with ONE bounding sphere only
        # (default: also with the minimum of its children's own bounds, rm_device.h: cull_union_children)
        self.cull_union_table = os.environ.get("RM_CULL_UNION_TABLE", "1") != "0"
        self.want_table = False        # set by the parent union for the child it emits next

    def off(self, *params):
        """Offset of the first parameter; the rest must follow contiguously."""
        base = self.offsets[id(params[0])]
        expect = base
        for p in params:
            if self.offsets[id(p)] != expect:
                raise ValueError("scene parameters of one node are not contiguous in named_parameters() "
                                 "order (shared nn.Parameter between fields of one node is not supported)")
            expect += p.numel()
        return base

    def push(self, n):
        self.depth += n
        self.max_depth = max(self.max_depth, self.depth)

    def pop(self, n):
        self.depth -= n

    def ins(self, op, off=0, a0=0, a1=0):
        self.code.append((op, off, a0, a1))


def _emit(node, em: _Emitter, n_params: int):
    kind = getattr(node, "_rm_kind", None)
    A = _abi
    if kind == "sphere":
        em.ins(A.OP_SPHERE, em.off(node.radius))
    elif kind == "box":
        em.ins(A.OP_BOX, em.off(node.halfsides))
    elif kind == "plane":
        em.ins(A.OP_PLANE)
    elif kind == "line":
        derived = n_params + em.next_line       # one of the slots reserved at the front of the derived block
        em.next_line += 6
        em.ins(A.OP_LINE, em.off(node.start, node.end, node.radius), derived)
    elif kind == "disk":
        em.ins(A.OP_DISK, em.off(node.radius))
    elif kind == "torus":
        em.ins(A.OP_TORUS, em.off(node.radius1, node.radius2))
    elif kind == "affine":
        off = em.off(node.translation, node.orientation)
        em.ins(A.OP_AFFINE_PUSH, off)
        em.push(_STACK_AFFINE)
        _emit(node.sdf, em, n_params)
        em.pop(_STACK_AFFINE)
        em.ins(A.OP_AFFINE_POP, off)
    elif kind in ("union", "smooth_union"):
        kids = list(node.sdfs)
        if not kids:
            raise ValueError("SDFUnion / SDFSmoothUnion needs at least one child")
        if kind == "smooth_union" and len(kids) >= 512:
            raise NotImplementedError("SDFSmoothUnion with 512 or more children: ATen's sum kernel switches to cascade "
                                      "levels there, which the kernels' summation order (aten_inner_sum) does not follow")
        base = em.n_slots
        smooth = kind == "smooth_union"
        em.n_slots += len(kids) + (1 if smooth else 0)     # smooth union: one more slot for the logsumexp value
        koff = em.off(node.blend_k) if smooth else 0
        # Exact culling inside a smooth union: a child whose term of the logsumexp is exactly +0.0f for every ray of the
        # wave (k (d_i - d_min) > 104) is skipped -- value, argmax and gradients unchanged (rm_device.h: lse_cull_mask).
        # The test runs once per evaluation, one child per lane, from a table of bounds (8 floats per child, 16-byte
        # aligned, filled in on the device); the skip bits live at the children's tape slots, hence slots < 64.
        lse_table = 0
        want_table, em.want_table = em.want_table, False
        cull_children = smooth and em.cull and em.cull_lse and em.cull_lse_min <= len(kids)
        want_table = want_table and em.cull_lse_min <= len(kids)        # ~45 instructions per test: not for a handful of children
        if smooth and em.cull and (cull_children or want_table) and len(kids) <= 64 and base + len(kids) <= 64:
            em.n_derived += (-(n_params + em.n_derived)) % 4
            lse_table = n_params + em.n_derived
            em.n_derived += 8 * len(kids)
        cull_children = cull_children and bool(lse_table)
        if smooth:
            em.ins(A.OP_SMOOTH_BEGIN, koff if lse_table else 0, lse_table,
                   ((int(cull_children) << 16) | (base << 8) | len(kids)) if lse_table else 0)
        else:
            em.ins(A.OP_UNION_BEGIN)
        em.push(_STACK_SMOOTH if smooth else _STACK_UNION)
        # Exact culling of min-union children.  min() does not care about the order its operands arrive in,
        # and the reverse pass finds the winner from the tape slots, which stay in the reference's child
        # order (ties -> first child); so the children may be EVALUATED in any order.  Children that cannot
        # be culled (cheap, or without a bounding sphere: the room shell, planes) go first, so that the
        # running minimum is already small when the expensive, boundable ones are tested against it.
        cullable = [(not smooth and em.cull and base + i < 64 and _cost(child) >= em.cull_min_cost
                     and _boundable(child)) for i, child in enumerate(kids)]
        order = list(range(len(kids)))
        if em.cull_reorder:
            order = [i for i in order if not cullable[i]] + [i for i in order if cullable[i]]
        for pos, i in enumerate(order):
            child = kids[i]
            cull_at = None
            if cullable[i] and pos > 0:       # never the first one evaluated: nothing to compare with yet
                cull_at = len(em.code)
                em.ins(A.OP_CULL_MIN, 0, n_params + em.n_derived, 0)      # aux1 patched below
                em.n_derived += 5      # {cx, cy, cz, K, slope}, filled in on the device
            lse_at = None
            if cull_children:
                lse_at = len(em.code)
                em.ins(A.OP_CULL_LSE, lse_table + 8 * i, base + i, 0)           # aux1 patched below
            if cull_at is not None and em.cull_union_table and getattr(child, "_rm_kind", None) == "smooth_union":
                em.want_table = True           # the child emits a bound table on its SMOOTH_BEGIN when its slots allow
            _emit(child, em, n_params)
            em.want_table = False
            if cull_at is not None:
                skip = len(em.code) - cull_at
                nxt = em.code[cull_at + 1]
                by_children = int(nxt[0] == A.OP_SMOOTH_BEGIN and nxt[2] != 0)     # also tested with the children's own bounds
                em.code[cull_at] = (A.OP_CULL_MIN, by_children, em.code[cull_at][2], (skip << 8) | (base + i))
                em.ins(A.OP_FOLD_MIN, 0, base + i, skip)
            elif lse_at is not None:
                skip = len(em.code) - lse_at
                em.code[lse_at] = (A.OP_CULL_LSE, lse_table + 8 * i, base + i, skip)
                em.ins(A.OP_FOLD_LSE, koff, base + i, skip)
            else:
                em.ins(A.OP_FOLD_LSE if smooth else A.OP_FOLD_MIN, koff, base + i)
        em.pop(_STACK_SMOOTH if smooth else _STACK_UNION)
        em.ins(A.OP_SMOOTH_END if smooth else A.OP_UNION_END, koff, base, len(kids))
    elif kind == "rounding":
        _emit(node.sdf, em, n_params)
        em.ins(A.OP_ROUND, em.off(node.rounding))
    elif kind == "onion":
        _emit(node.sdf, em, n_params)
        slot = em.n_slots
        em.n_slots += 1
        em.ins(A.OP_ONION, em.off(node.radius), slot)
    else:
        raise TypeError(
            f"{type(node).__name__} is not a ray_marching_amd SDF node; only the node types of "
            "ray_marching_amd.scene (the reference's 6 primitives and 5 combinators) can be compiled "
            "for the HIP kernels")


def compile_scene(module: nn.Module) -> CompiledScene:
    """Lower an SDF module tree.  Pure host logic (no GPU needed)."""
    names, leaves, offsets, table = [], [], [], {}
    cursor = 0
    for name, p in module.named_parameters():
        names.append(name)
        leaves.append(p)
        offsets.append(cursor)
        table[id(p)] = cursor
        cursor += p.numel()
    n_params = cursor
    em = _Emitter(table)
    # derived block = [capsule constants of every SDFLine (gradients flow through them) | cull bounds, bound tables]
    em.n_grad_derived = em.n_derived = 6 * sum(1 for m in module.modules() if getattr(m, "_rm_kind", None) == "line")
    _emit(module, em, n_params)
    if em.next_line != em.n_grad_derived:
        raise ValueError("an SDF node instance appears more than once in the scene tree (shared sub-modules are not supported)")
    program = np.asarray(em.code, dtype=np.int32).reshape(-1, 4)
    rc = _abi.lib.rm_validate_program(program.ctypes.data, program.shape[0], n_params, em.n_derived,
                                      em.max_depth, em.n_slots)
    _abi.check(rc, "rm_validate_program")
    signature = (tuple(map(tuple, program.tolist())), n_params, em.n_derived, em.max_depth, em.n_slots, em.n_grad_derived)
    return CompiledScene(program=program, leaves=leaves, leaf_names=names, leaf_offsets=offsets,
                         n_params=n_params, n_derived=em.n_derived, n_grad_derived=em.n_grad_derived, stack_floats=em.max_depth,
                         n_slots=em.n_slots, signature=signature)


def structure_key(module: nn.Module):
    """Cheap fingerprint that changes when the tree topology or parameter identity changes: the ids of
    every sub-module and parameter, walked through ``_modules`` / ``_parameters`` directly
    (``module.parameters()`` builds dotted name strings on every call: 25 us for a 7-node scene)."""
    out = []
    stack = [module]
    while stack:
        m = stack.pop()
        out.append(id(m))
        for p in m._parameters.values():
            out.append(id(p))
        stack.extend(m._modules.values())
    return tuple(out)


_compiled = weakref.WeakKeyDictionary()      # module -> (structure key, CompiledScene); NOT in the module's own
                                              # state: copy.deepcopy / torch.save of a scene or RenderLoop must work


def compiled_for(module: nn.Module) -> CompiledScene:
    """Compile once per (module instance, topology)."""
    key = structure_key(module)
    cache = _compiled.get(module)
    if cache is None or cache[0] != key:
        cache = (key, compile_scene(module))
        _compiled[module] = cache
    return cache[1]
