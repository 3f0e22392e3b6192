import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The GPU tests pipeline frames over several HIP streams; the package no longer exports GPU_MAX_HW_QUEUES when it is imported (round 5), the
# integrator calls configure_runtime() before the first HIP call - for the test process that is here.
import gaussianmesh_amd  # noqa: E402

gaussianmesh_amd.configure_runtime()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
