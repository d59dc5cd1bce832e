import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    ge.load_package()
    import dxpbrt_amd.layouts as L
    import dxpbrt_amd.scenes as S

    class P:
        layouts = L
        scenes = S
    return P


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/): the checker, never the thing under test on the GPU side."""
    mod = ge.load_oracle()
    mod.lib()
    return mod


@pytest.fixture(scope="session")
def ptamd():
    ge.load_package()
    import dxpbrt_amd.ptamd as P
    return P


@pytest.fixture(scope="session")
def gpu(ptamd):
    """A device context on cuda:0. Fails (not skips) when the HIP library is missing."""
    ptamd.load_library()
    ctx = ptamd.DeviceContext(0)
    yield ctx
    ctx.close()
