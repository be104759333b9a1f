import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than a few seconds on CPU")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Build the in-tree libraries once if they are missing (hipcc cross-compiles gfx950 without a GPU)."""
    from swarm_simulator_amd import _abi as A
    need = [os.path.join(A.LIB_DIR, "librbp_hip.so"), os.path.join(A.LIB_DIR, "librbp_host.so"),
            os.path.join(ROOT, "oracle", "_build", "librbp_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__ as g
        g.build()
