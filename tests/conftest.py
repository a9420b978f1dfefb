import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement (test infrastructure).  Built on demand with g++."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def gpu_ctx():
    import lili_om_amd as L
    ctx = L.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture
def launch_by_launch(gpu_ctx):
    """The launch-per-stage iteration loop (the default).  Option persistent_iterate = 1 runs small scans inside ONE persistent launch
    (k_iterate_coop) whose Gram sums are partitioned per workgroup of 256 / L queries: poses then differ in the last bits.  Tests that assert
    BIT-identity between launch structures of the launch-per-stage loop pin the option, whatever the default is."""
    gpu_ctx.set_option("persistent_iterate", 0)
    yield gpu_ctx
    gpu_ctx.set_option("persistent_iterate", 0)
