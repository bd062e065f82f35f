import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Builds oracle/build/libgv_oracle.so on first use."""
    from oracle import oracle_py
    oracle_py.load()
    return oracle_py


@pytest.fixture(scope="session")
def gpu():
    """One libgarden_vis context on cuda:0. Fails loudly (no fallback) if the HIP library or device is missing."""
    from garden_amd.lib import GpuVisibility
    vis = GpuVisibility(device=0, profile_events=True)
    yield vis
    vis.close()



@pytest.fixture(scope="session")
def gpu_slot_order():
    """Context with GV_CONFIG_KEEP_SLOT_ORDER (mirror in pool-slot order, no spatial permutation)."""
    from garden_amd.lib import GpuVisibility
    vis = GpuVisibility(device=0, keep_slot_order=True)
    yield vis
    vis.close()
