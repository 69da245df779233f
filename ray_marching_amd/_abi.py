"""ctypes binding of librm_hip.so (C ABI declared in include/rm_abi.h).

The library is the product; there is no Python/PyTorch fallback.  If the shared
object is missing or does not export the ABI this module raises at import time.
"""
from __future__ import annotations

import ctypes as C
import os

# torch FIRST: it carries its own libamdhip64.so (soname libamdhip64.so.7, loaded by path).  If librm_hip.so
# is opened before it, the loader resolves our libamdhip64.so.7 to /opt/rocm/lib's copy, torch then maps its
# bundled one next to it, and the process holds two HIP runtimes -- ours then fails every launch with "no
# ROCm-capable device is detected" (seen on the GPU box with build() followed by smoke() in one process).
import torch  # noqa: F401,E402

ABI_VERSION = 13
_HERE = os.path.dirname(os.path.abspath(__file__))
from ._build import LIB_PATH  # noqa: E402  (ray_marching_amd/lib/librm_hip.so, or under RM_LIB_DIR)

# opcodes (include/rm_abi.h)
OP_SPHERE, OP_BOX, OP_PLANE, OP_LINE, OP_DISK, OP_TORUS = 1, 2, 3, 4, 5, 6
OP_AFFINE_PUSH, OP_AFFINE_POP = 7, 8
OP_UNION_BEGIN, OP_FOLD_MIN, OP_UNION_END = 9, 10, 11
OP_SMOOTH_BEGIN, OP_FOLD_LSE, OP_SMOOTH_END = 12, 13, 14
OP_ROUND, OP_ONION = 15, 16
OP_CULL_MIN = 17
OP_CULL_LSE = 18

FLAG_EARLY_OUT, FLAG_TILE8X8, FLAG_DYNAMIC_TILES, FLAG_REGEN, FLAG_ORDER_PER_RAY = 1, 2, 4, 8, 16
ORDER_ONE_BLOCK, ORDER_SCRATCH_INTS = 131072, 8192
DTYPE_F32, DTYPE_F16, DTYPE_F64, DTYPE_RGBA_F32 = 0, 1, 2, 3      # RM_DTYPE_*
_DTYPES = {torch.float32: DTYPE_F32, torch.float16: DTYPE_F16, torch.float64: DTYPE_F64}


def dtype_code(dtype) -> int:
    """RM_DTYPE_* of a torch dtype (KeyError for anything the kernels cannot read or write)."""
    return _DTYPES[dtype]
WORK_WORDS = 64 + 64 * 32 + 8 * 4 * 32 + 64 * 32   # RM_WORK_WORDS (min/max words, tile queues, parking counters, min/max slots)
CAMERA_BWD_BLOCKS = 256     # RM_CAMERA_BWD_BLOCKS
NORM_BWD_BLOCKS = 1024      # RM_NORM_BWD_BLOCKS
MODES = ("lambertian", "distance", "proximity", "vignette", "normal", "laplacian", "tangent", "spin")


class RmScene(C.Structure):
    _fields_ = [("program", C.c_void_p), ("params", C.c_void_p), ("param_refs", C.c_void_p), ("n_instr", C.c_int32),
                ("n_params", C.c_int32), ("n_derived", C.c_int32), ("stack_floats", C.c_int32),
                ("n_slots", C.c_int32), ("n_grad_derived", C.c_int32), ("block", C.c_void_p), ("block_out", C.c_void_p),
                ("block_cache", C.c_void_p)]


class RmCamera(C.Structure):
    _fields_ = [("ray_positions", C.c_void_p), ("ray_directions", C.c_void_p),
                ("num_cameras", C.c_int32), ("height", C.c_int32), ("width", C.c_int32), ("dtype", C.c_int32)]


class RmTetra(C.Structure):
    _fields_ = [("offsets", C.c_float * 12), ("inverse", C.c_float * 9), ("lap_scale", C.c_float)]


class RmError(RuntimeError):
    pass


_P = C.c_void_p
_SIGNATURES = {
    "rm_abi_version": (C.c_int, []),
    "rm_last_error": (C.c_char_p, []),
    "rm_grad_partials_floats": (C.c_int64, [C.POINTER(RmScene), C.c_int64]),
    "rm_validate_program": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rm_sdf_forward": (C.c_int, [C.POINTER(RmScene), _P, _P, C.c_int64, C.c_int32, _P]),
    "rm_sdf_backward": (C.c_int, [C.POINTER(RmScene), _P, _P, _P, _P, _P, C.c_int64, _P]),
    "rm_march_forward": (C.c_int, [C.POINTER(RmScene), _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "rm_march_backward": (C.c_int, [C.POINTER(RmScene), _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, _P]),
    "rm_normals_forward": (C.c_int, [C.POINTER(RmScene), C.POINTER(RmTetra), _P, _P, _P, C.c_int64, C.c_int32, _P]),
    "rm_normals_backward": (C.c_int, [C.POINTER(RmScene), C.POINTER(RmTetra), _P, _P, _P, _P, _P, _P, C.c_int64, _P]),
    "rm_camera_forward": (C.c_int, [C.POINTER(RmCamera), _P, _P, _P, _P, _P, _P]),
    "rm_render_forward": (C.c_int, [C.POINTER(RmScene), C.POINTER(RmCamera), C.POINTER(RmTetra), _P, _P,
                                    _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, _P]),
    "rm_render_traj_floats": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rm_park_floats": (C.c_int64, [C.c_int64]),
    "rm_wave_tiles": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rm_tile_order_from_cost": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P, _P]),
    "rm_tile_score_from_ray_cost": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "rm_minmax_init": (C.c_int, [_P, _P]),
    "rm_minmax_init_many": (C.c_int, [_P, C.c_int32, _P]),
    "rm_minmax_decode": (C.c_int, [_P, _P, _P]),
    "rm_minmax_encode": (C.c_int, [_P, _P, _P]),
    "rm_shade_finish": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, C.c_int32, C.c_int32, _P]),
    "rm_shade_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, _P, _P, C.c_int32, C.c_int32,
                                   C.c_int32, C.c_int32, C.c_int64, C.c_int64, _P]),
    "rm_shade_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int64, _P]),
    "rm_render_backward": (C.c_int, [C.POINTER(RmScene), C.POINTER(RmCamera), C.POINTER(RmTetra), _P, _P,
                                     _P, _P, _P, _P, _P, _P, _P, _P, _P, _P,
                                     _P, _P, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int64, _P]),
    "rm_sum_rows": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P]),
    "rm_shade_norm_backward": (C.c_int, [_P, _P, _P, C.c_int32, _P, _P, C.c_int64, _P]),
    "rm_bwd_hard_floats": (C.c_int64, [C.c_int64, C.c_int32]),
    "rm_camera_backward": (C.c_int, [C.POINTER(RmCamera), _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def _load():
    from . import _build
    if _build.library_is_stale():
        # fresh checkout or edited kernels: build now if a compiler is here, otherwise fail loudly
        if _build.hipcc() is None:
            raise ImportError(
                f"{LIB_PATH} is missing or older than its sources and hipcc is not available: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).  "
                "ray_marching_amd has no CPU or PyTorch fallback.")
        _build.build_library()
    return bind(C.CDLL(LIB_PATH))


def bind(lib):
    """Attach the ABI signatures to a loaded library (generic or per-scene specialised)."""
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI symbol missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    got = lib.rm_abi_version()
    if got != ABI_VERSION:
        raise ImportError(f"{lib._name}: ABI version {got}, expected {ABI_VERSION}: rebuild")
    return lib


lib = _load()
_fast = None


def fast_lib():
    """The opt-in 'fast' arithmetic build of the generic library (built on first use)."""
    global _fast
    if _fast is None:
        from . import _build
        _fast = bind(C.CDLL(_build.build_library(precision="fast")))
    return _fast


def generic_lib(precision: str = "exact"):
    return lib if precision == "exact" else fast_lib()


def check(code: int, what: str, from_lib=None):
    if code != 0:
        raise RmError(f"{what} failed ({code}): {(from_lib or lib).rm_last_error().decode()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
