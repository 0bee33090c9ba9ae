import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "metagenomic-deepfri_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_finish(session):
    """GPU runs: import torch BEFORE any test lets libmdfri_hip initialise HIP.  The other order works (the library pre-loads
    torch's bundled HIP runtime, tests/test_gpu_gcn.py::test_library_first_then_torch_share_one_hip_runtime) but makes the HIP
    runtime digest every fat binary of libtorch_hip eagerly at `import torch` -- seconds with a warm page cache, minutes the
    first time on a fresh box (measured on this pool: 203 s, then 2.5 s)."""
    if any(item.get_closest_marker("gpu") is not None for item in session.items):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass


@pytest.fixture(scope="session")
def cmap_golden():
    return np.load(os.path.join(GOLDEN, "cmap_golden.npz"))


@pytest.fixture(scope="session")
def gcn_golden():
    return np.load(os.path.join(GOLDEN, "gcn_golden.npz"))


@pytest.fixture(scope="session")
def nw_golden():
    return np.load(os.path.join(GOLDEN, "nw_golden.npz"))


def gstr(a) -> str:
    return bytes(np.asarray(a, dtype=np.uint8)).decode()
