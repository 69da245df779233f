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
# RM_LIB_DIR: build into / load from another directory (variant builds of tests and A/B probes: the product library
# under ray_marching_amd/lib/ is then never replaced)
LIBDIR = os.environ.get("RM_LIB_DIR") or os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIBDIR, "librm_hip.so")
SOURCES = [os.path.join(CSRC, f) for f in ("rm_abi.hip", "rm_kernels.h", "rm_device.h", "rm_math.h")] + \
    [os.path.join(os.path.dirname(_HERE), "include", "rm_abi.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-shared", "-fPIC", "-std=c++17"]
# opt-in "fast" arithmetic (RenderLoop(precision="fast")): 1-ulp v_sqrt_f32, reciprocal normalise, FMA
# contraction.  Still fp32 throughout; no longer bit-identical with the reference's CPU op stream.
HIPCC_FLAGS_FAST = ["--offload-arch=gfx950", "-O3", "-ffp-contract=fast", "-DRM_FAST_MATH", "-shared", "-fPIC",
                    "-std=c++17"]
LIB_PATH_FAST = os.path.join(LIBDIR, "librm_hip_fast.so")


def extra_flags():
    """Experiment knob: RM_HIPCC_EXTRA="-DRM_..." adds compile flags (A/B probes, profiles/ab_probe.py); the
    flags are part of every library's staleness / cache key, so variants never mix."""
    return os.environ.get("RM_HIPCC_EXTRA", "").split()


def variant(precision: str):
    if precision == "exact":
        return LIB_PATH, HIPCC_FLAGS + extra_flags()
    if precision == "fast":
        return LIB_PATH_FAST, HIPCC_FLAGS_FAST + extra_flags()
    raise ValueError(f"precision must be 'exact' or 'fast', not {precision!r}")


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
    h.update(" ".join(extra_flags()).encode())
    return h.hexdigest()


def library_is_stale(precision: str = "exact") -> bool:
    """Content-hash staleness (file times do not survive the repo snapshot to the GPU box)."""
    path, _ = variant(precision)
    stamp = path + ".srchash"
    if not (os.path.isfile(path) and os.path.isfile(stamp)):
        return True
    with open(stamp) as f:
        return f.read().strip() != sources_hash()


def build_library(force: bool = False, precision: str = "exact") -> str:
    """hipcc rm_abi.hip -> ray_marching_amd/lib/librm_hip[_fast].so (in-tree so it travels to the GPU box)."""
    path, flags = variant(precision)
    if not force and not library_is_stale(precision):
        return path
    cc = hipcc()
    if cc is None:
        raise RuntimeError(f"hipcc not found: cannot build {os.path.basename(path)}")
    os.makedirs(LIBDIR, exist_ok=True)
    tmp = path + f".tmp{os.getpid()}"
    subprocess.run([cc, *flags, os.path.join(CSRC, "rm_abi.hip"), "-o", tmp], check=True, cwd=CSRC)
    os.replace(tmp, path)
    with open(path + ".srchash", "w") as f:
        f.write(sources_hash())
    return path
