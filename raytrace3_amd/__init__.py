"""raytrace3_amd — MI355X-native SPPM hot path (eye trace + photon trace/gather).

The compute lives in csrc/libsppm_hip.so (hand-written HIP for gfx950) behind the
C-ABI of include/sppm_abi.h; this package is the Python host side used by the
tests and bench.py.  Nothing here falls back to a CPU implementation.
"""
from . import abi, scenes  # noqa: F401
from .abi import Film, SppmContext, SppmError  # noqa: F401
