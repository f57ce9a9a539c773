import os
import sys

# several replicas of ONE process that wait for each other inside kernels (test_one_kernel_exchange_among_several_replicas) need a
# hardware queue each: HIP maps streams onto 4 by default, a fifth stream shares one -- and a waiting kernel then blocks its peer
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# a peer that never shows up: the exchange kernel gives up after this long (the library's own default is one minute)
os.environ.setdefault("SMARTIES_HIP_XCHG_TIMEOUT_MS", "30000")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


@pytest.fixture(scope="session")
def hip_api():
    """The product library (hand-written HIP behind the hl_* C-ABI).  No fallback."""
    from smarties_amd import load_hip
    return load_hip()
