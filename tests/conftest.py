import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _oracle_path():
    return os.path.join(ROOT, "oracle", "liboracle.so")


@pytest.fixture(scope="session")
def orc():
    """ctypes binding of the CPU oracle (test infrastructure), prefix orc_."""
    import web_audio_api_rs_amd as waa

    path = _oracle_path()
    src = os.path.join(ROOT, "oracle", "waa_oracle.c")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    lib = ctypes.CDLL(path)
    return waa.bind(lib, "orc_")


@pytest.fixture(scope="session")
def orc_lib(orc):
    return orc.lib


@pytest.fixture(scope="session")
def hip():
    """ctypes binding of the product library; fails loudly if it is not built."""
    import web_audio_api_rs_amd as waa

    return waa.default_binding()


@pytest.fixture(params=["orc", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    """Backend under test: the CPU oracle (pins the oracle) or the HIP library (-m gpu)."""
    return request.getfixturevalue(request.param)
