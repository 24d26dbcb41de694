"""web-audio-api-rs_amd — MI355X-native batched offline render engine (hot path only).

Host-side mirror of the reference control API (api.py) over the C-ABI shared library
``csrc/libwaa_hip.so`` (include/waa_hip.h).  The HIP library is the only compute path:
there is no CPU fallback, and importing the default binding fails loudly if the
library has not been built (``python -c "import __graft_entry__ as g; g.build()"``).
"""
import ctypes
import os

from .api import *  # noqa: F401,F403
from .api import Binding, bind
from .mixed import bucket_report, render_contexts, shape_key  # noqa: F401  (contexts of different shapes -> batches)

_HERE = globals().get("_WAA_PKG_DIR") or os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libwaa_hip.so")
# the same sources built with -DWAA_MEASURE: the only library in which the A/B / debugging / measurement switches
# (measure_switch(), csrc/waa_internal.hpp) are read.  Tests marked `measure` and the tools under tools/ load it; nothing
# that reports a number or ships does.
MEASURE_LIB_PATH = os.path.join(_HERE, "csrc", "libwaa_hip_measure.so")
_default = None
_measure = None


def _load(path) -> Binding:
    if not os.path.exists(path):
        raise ImportError(
            f"{path} not found: the HIP extension is not built. Run __graft_entry__.build() "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # RTLD_GLOBAL: the HIP runtime this library brings in must be the ONE runtime of the process — torch, imported later,
    # otherwise initialises a second copy that sees no devices (measured: "no ROCm-capable device is detected").  The two
    # builds of the library export the same symbols; they are linked -Bsymbolic, so each calls into itself.
    return bind(ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL), "waa_")


def default_binding() -> Binding:
    """ctypes binding of the HIP library.  Raises if it is missing — never falls back.  (WAA_USE_MEASURE_LIB=1 in the
    environment makes this the measurement build: how tools/ab_*.py flip switches under bench.py's workload builders.)"""
    global _default
    if os.environ.get("WAA_USE_MEASURE_LIB") == "1":
        return measure_binding()
    if _default is None:
        _default = _load(LIB_PATH)
    return _default


def measure_binding() -> Binding:
    """the measurement build (libwaa_hip_measure.so): same code, measurement switches alive"""
    global _measure
    if _measure is None:
        _measure = _load(MEASURE_LIB_PATH)
    return _measure
