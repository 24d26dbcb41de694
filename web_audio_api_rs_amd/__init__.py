"""Importable alias of the package directory ``web-audio-api-rs_amd/`` (a hyphen is not a
valid Python identifier, so this shim points the package path at it)."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "web-audio-api-rs_amd")
__path__ = [_real]
_WAA_PKG_DIR = _real
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
