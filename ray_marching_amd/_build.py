"""Building librm_hip.so (hipcc, gfx950 only).  Used by __graft_entry__.build() and, on a fresh
checkout, by the first ``import ray_marching_amd`` when hipcc is available.  There is still no fallback:
without the library (and without hipcc to make it) the import fails."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIBDIR, "librm_hip.so")
SOURCES = [os.path.join(CSRC, f) for f in ("rm_abi.hip", "rm_kernels.h", "rm_device.h")] + \
    [os.path.join(os.path.dirname(_HERE), "include", "rm_abi.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC", "-std=c++17"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    return shutil.which("hipcc")


def sources_hash() -> str:
    h = hashlib.sha1()
    for s in SOURCES:
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def library_is_stale() -> bool:
    """Content-hash staleness (file times do not survive the repo snapshot to the GPU box)."""
    stamp = LIB_PATH + ".srchash"
    if not (os.path.isfile(LIB_PATH) and os.path.isfile(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != sources_hash()


def build_library(force: bool = False) -> str:
    """hipcc rm_abi.hip -> ray_marching_amd/lib/librm_hip.so (in-tree so it travels to the GPU box)."""
    if not force and not library_is_stale():
        return LIB_PATH
    cc = hipcc()
    if cc is None:
        raise RuntimeError("hipcc not found: cannot build librm_hip.so")
    os.makedirs(LIBDIR, exist_ok=True)
    tmp = LIB_PATH + f".tmp{os.getpid()}"
    subprocess.run([cc, *HIPCC_FLAGS, os.path.join(CSRC, "rm_abi.hip"), "-o", tmp], check=True, cwd=CSRC)
    os.replace(tmp, LIB_PATH)
    with open(LIB_PATH + ".srchash", "w") as f:
        f.write(sources_hash())
    return LIB_PATH
