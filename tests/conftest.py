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


@pytest.fixture(scope="session")
def cmap_golden():
    return np.load(os.path.join(GOLDEN, "cmap_golden.npz"))


@pytest.fixture(scope="session")
def gcn_golden():
    return np.load(os.path.join(GOLDEN, "gcn_golden.npz"))


def gstr(a) -> str:
    return bytes(np.asarray(a, dtype=np.uint8)).decode()
