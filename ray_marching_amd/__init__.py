"""ray_marching_amd -- MI355X-native sphere-tracing renderer.

Drop-in for the hot path of kyle-rosa/ray_marching (RenderLoop.forward and the
nn.Module SDF interface).  All arithmetic runs in hand-written HIP kernels
(csrc/) behind the C ABI of include/rm_abi.h; importing the package loads
lib/librm_hip.so and fails loudly if it is missing.  There is no CPU fallback.
"""
from . import _abi  # noqa: F401  (loads the shared library; raises if absent)

__all__ = ["_abi"]
