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

_HERE = globals().get("_WAA_PKG_DIR") or os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libwaa_hip.so")
_default = None


def default_binding() -> Binding:
    """ctypes binding of the HIP library.  Raises if it is missing — never falls back."""
    global _default
    if _default is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _default = bind(lib, "waa_")
    return _default
