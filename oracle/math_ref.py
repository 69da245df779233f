"""ctypes access to oracle/rm_math_ref.c (TEST INFRASTRUCTURE ONLY -- see that file's header).

Only tests/, __graft_entry__ and oracle/gen_math_golden.py import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "librm_math_ref.so")
FN = {"exp": 0, "log": 1, "pow_gamma": 2, "atan2": 3, "sqrt": 4}
# sqrt: the device function is specified for |x| >= 2^-95 (below, v_sqrt_f32 flushes denormal inputs and the FMA
# residuals underflow): blocks 0x00-0x0f and 0x80-0x8f are not compared
CHECKED_BLOCKS = {"sqrt": [[0x10, 0x80], [0x90, 0x100]]}
BLOCK = 1 << 24
GAMMA = float(torch.tensor(1 / 2.33, dtype=torch.float32))   # the shader's exponent as ATen sees it


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rm_math_ref.c")
    if force or not os.path.isfile(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.rm_sweep_block.restype = C.c_uint64
        _lib.rm_sweep_block.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int64)]
        _lib.rm_sweep_inputs.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        for name in ("ref_expf_v", "ref_logf_v", "ref_sqrtf_v"):
            getattr(_lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        for name in ("ref_powf_v", "ref_atan2f_v"):
            getattr(_lib, name).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    return _lib


def _unary(name, x):
    x = x.detach().float().contiguous()
    out = torch.empty_like(x)
    getattr(lib(), name)(x.data_ptr(), out.data_ptr(), x.numel())
    return out


def _binary(name, a, b):
    a, b = torch.broadcast_tensors(a.detach().float(), b.detach().float())
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    getattr(lib(), name)(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel())
    return out


def expf(x): return _unary("ref_expf_v", x)
def logf(x): return _unary("ref_logf_v", x)
def sqrtf(x): return _unary("ref_sqrtf_v", x)
def powf(x, y): return _binary("ref_powf_v", x, torch.as_tensor(y, dtype=torch.float32))
def atan2f(y, x): return _binary("ref_atan2f_v", y, x)


def sweep_inputs(fn: str, block: int):
    a = torch.empty(BLOCK, dtype=torch.float32)
    b = torch.empty(BLOCK, dtype=torch.float32) if fn == "atan2" else None
    lib().rm_sweep_inputs(FN[fn], block, a.data_ptr(), None if b is None else b.data_ptr())
    return a, b


def torch_eval(fn: str, a, b):
    """The reference's call for this function, on this host's torch CPU."""
    if fn == "exp":
        return torch.exp(a)
    if fn == "log":
        return torch.log(a)
    if fn == "sqrt":
        return a.pow(1 / 2)              # shader.py:116 (ATen dispatches exponent 0.5 to its sqrt kernel = MKL vsSqrt)
    if fn == "pow_gamma":
        return a.pow(1 / 2.33)           # shader.py:37
    return torch.atan2(a, b)             # shader.py:99 (imag, real)


def sweep_block(fn: str, block: int, other=None):
    """-> (checksum of the restatement, checksum of `other`, n inputs that differ, max ulp distance)."""
    osum, nd, mu = C.c_uint64(0), C.c_int64(0), C.c_int64(0)
    s = lib().rm_sweep_block(FN[fn], block, None if other is None else other.data_ptr(), C.byref(osum), C.byref(nd),
                             C.byref(mu))
    return int(s), int(osum.value), int(nd.value), int(mu.value)
