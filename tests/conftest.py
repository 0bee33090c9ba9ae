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
def cmap_special_golden():
    """tests/golden/make_special_golden.py: subnormal / overflowing / signed-zero / NaN / inf coordinates and on-threshold pairs through the
    compiled reference (D as bit patterns, the map per threshold, the aligned map)."""
    return np.load(os.path.join(GOLDEN, "cmap_special_golden.npz"))


def special_cases(g):
    """(name, X, D bits, [(threshold, map)], {gen: aligned map}) per coordinate set of cmap_special_golden.npz"""
    thr = [float(t) for t in g["thresholds"]]
    for name in [str(x) for x in g["index/sets"]]:
        X = np.ascontiguousarray(g[name + "/X"])
        n = X.shape[0]
        maps = [(t, np.unpackbits(g[f"{name}/cmap_bits/{k}"], axis=1)[:, :n].astype(np.int32)) for k, t in enumerate(thr)]
        aligned = {gen: np.unpackbits(g[f"{name}/aligned_bits/gen{gen}"], axis=1)[:, :n].astype(np.int32) for gen in (0, 2)}
        yield name, X, g[name + "/D_bits"], maps, aligned


def same_float_bits(got, want_bits):
    """float32 array == stored bit patterns, cell by cell; where the reference has a NaN only NaN-ness is compared (its sign and payload are
    platform-specific: x86 SSE produces the negative default NaN, gfx950 the positive one; every later use is `D < thr`, false for any NaN)"""
    want = np.asarray(want_bits, dtype=np.uint32).view(np.float32)
    nan = np.isnan(want)
    return got.dtype == np.float32 and got.shape == want.shape and np.array_equal(np.isnan(got), nan) and \
        np.array_equal(got.view(np.uint32)[~nan], np.asarray(want_bits, dtype=np.uint32)[~nan])


@pytest.fixture(scope="session")
def gcn_golden():
    return np.load(os.path.join(GOLDEN, "gcn_golden.npz"))


@pytest.fixture(scope="session")
def nw_golden():
    return np.load(os.path.join(GOLDEN, "nw_golden.npz"))


def gstr(a) -> str:
    return bytes(np.asarray(a, dtype=np.uint8)).decode()
