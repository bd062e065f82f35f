import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """A fresh checkout has no binaries (they are git-ignored): build the HIP library, the oracle and the C++ test
    driver once, exactly as __graft_entry__.build() does (hipcc cross-compiles without a GPU)."""
    lib = os.path.join(ROOT, "garden_amd", "lib", "libgarden_vis.so")
    orc = os.path.join(ROOT, "oracle", "build", "libgv_oracle.so")
    tick = os.path.join(ROOT, "tests", "cpp", "build", "headless_tick")
    ranks = os.path.join(ROOT, "tests", "cpp", "build", "exchange_ranks")
    stub = os.path.join(ROOT, "tests", "cpp", "build", "librccl_stub.so")
    if not (os.path.exists(lib) and os.path.exists(orc) and os.path.exists(tick) and os.path.exists(ranks) and os.path.exists(stub)):
        import __graft_entry__
        __graft_entry__.build()


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


@pytest.fixture(scope="session")
def gpu_bounds():
    """Context with GV_CONFIG_BLOCK_BOUNDS (workgroup boxes; conservative block-level frustum rejection)."""
    from garden_amd.lib import GpuVisibility
    vis = GpuVisibility(device=0, block_bounds=True)
    yield vis
    vis.close()


@pytest.fixture(scope="session")
def gpu_linear():
    """Context with GV_CONFIG_LINEAR_SCAN: every workgroup reads its streams whatever the pool's size (no block bounds)."""
    from garden_amd.lib import GpuVisibility
    vis = GpuVisibility(device=0, linear_scan=True)
    yield vis
    vis.close()
