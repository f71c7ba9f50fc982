import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library reads its tuning / A-B knobs (PBRT_HIP_MIN_WALKERS, PBRT_HIP_COLLAPSE, ...) only behind this switch
os.environ.setdefault("PBRT_HIP_DEBUG_KNOBS", "1")
# the device builder optimises trees of >= 1024 triangles by re-insertion (smaller ones gain nothing); the suite lowers that to 8 so that
# its small scenes (36 .. 1014 triangles, the random ones of 0 .. 300) exercise the pass too -- "gpu-plain" covers the tree as built
os.environ.setdefault("PBRT_HIP_REINSERT_MIN_TRIS", "8")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the HIP library and the oracle once per session if they are missing (hipcc
    cross-compiles on CPU; on the GPU box the prebuilt .so files travel with the snapshot)."""
    from pbrt_amd.build import build_hip
    build_hip()
    from oracle import binding
    binding.build()


@pytest.fixture(scope="session")
def oracle():
    from oracle import binding
    return binding


@pytest.fixture(scope="session")
def gpu():
    import pbrt_amd
    if pbrt_amd.device_count() < 1:
        pytest.fail("a gpu-marked test ran without a HIP device: pbrt_amd has no CPU fallback")
    return pbrt_amd
