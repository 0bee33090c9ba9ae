#!/usr/bin/env python3
"""Special-value fixtures for a1 / a2 / a4 from the REAL reference kernels (oracle/_ref = /root/reference/mDeepFRI/contact_map_utils.pyx
compiled where it lies by oracle/build_ref.py; the thresholding is the numpy expression of bio_utils.py:214-220 executed here).

    python oracle/build_ref.py && python tests/golden/make_special_golden.py      # dev container only -> cmap_special_golden.npz

What the random goldens never reach: squared distances that are SUBNORMAL in float32 (points 1e-20 apart), that OVERFLOW to
inf (coordinates of +-3e19), -0.0 coordinates, 1e18-scale magnitudes, a NaN row and an inf row, pairs at exactly the threshold distance, and
thresholds one float32 step either side of 6 A, tiny (1e-19 A) and huge (3e19 A: its square overflows float32).  200 residues per set, the
special rows far apart in index, so that the batched kernels meet them in their diagonal chunks, their packed two-rows-per-instruction
chunks and across 64-column chunk borders.  Stored: coordinates, D as BIT PATTERNS (uint32), the int32 map per threshold (bit-packed), and
the aligned map of the identity alignment for generated_contacts 0 and 2 at 6 A.  Data only; no reference source."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

import build_ref  # noqa: E402
from mdfri_testkit import synthetic  # noqa: E402

THRESHOLDS = [6.0, float(np.nextafter(np.float32(6.0), np.float32(7.0))), float(np.nextafter(np.float32(6.0), np.float32(5.0))), 1e-19, 3e19, 0.0]


def sets():
    rng = np.random.default_rng(2026)
    n = 200
    out = {}
    # A: a cluster of points 1e-20 / 3e-23 apart (subnormal d^2, some rounding to the smallest subnormal or to zero), signed zeros, and pairs at
    # exactly 6 A (axis-parallel: 36.0 exactly; 3.6 / 4.8: the float32 roundings decide which side of 36 the sum lands)
    X = synthetic.random_walk_coords(rng, n).astype(np.float32) + np.float32(50.0)
    tiny = np.array([[0, 0, 0], [1e-20, 0, 0], [2e-20, 1e-20, 0], [3e-23, 0, 0], [0, 3e-23, 2e-23], [1e-19, -1e-19, 1e-19]], dtype=np.float32)
    for k, i in enumerate((0, 1, 2, 150, 151, 199)):
        X[i] = tiny[k]
    X[30] = np.array([-0.0, -0.0, -0.0], dtype=np.float32)
    X[170] = np.array([0.0, -0.0, 0.0], dtype=np.float32)
    X[40], X[41], X[180], X[100] = (100, 0, 0), (106, 0, 0), (103.6, 4.8, 0), (100, 0, 6)
    out["tiny_zero_edge"] = X
    # B: magnitudes whose squared differences overflow (3e19, 1.5e19) or nearly do (1e18: d^2 ~ 1e36 .. 1.2e37), a NaN row, an inf row
    X = synthetic.random_walk_coords(rng, n).astype(np.float32)
    X[10], X[140] = (3e19, 0, 0), (-3e19, 1.0, 2.0)
    X[11], X[141] = (1.5e19, 1.5e19, 0), (-1.5e19, 0, 1.5e19)
    X[12], X[142] = (1e18, -1e18, 1e18), (-1e18, 1e18, -1e18)
    X[20] = (np.nan, 1.0, 2.0)
    X[160] = (np.inf, -np.inf, 0.0)
    X[161] = (np.inf, 0.0, 0.0)
    out["huge_nan_inf"] = X
    return out


def main():
    ref = build_ref.load()
    if ref is None:
        build_ref.build()
        ref = build_ref.load()
    assert ref is not None, "reference build unavailable"
    g = {"index/sets": np.array(list(sets())), "thresholds": np.array(THRESHOLDS, dtype=np.float64)}
    for name, X in sets().items():
        X = np.ascontiguousarray(X, dtype=np.float32)
        D = ref.pairwise_sqeuclidean(X)
        g[f"{name}/X"] = X
        g[f"{name}/D_bits"] = D.view(np.uint32)
        for k, thr in enumerate(THRESHOLDS):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")       # (3e19 ** 2 overflows the float32 the comparison is made in: numpy warns and compares with inf)
                cm = (D < thr ** 2).astype(np.int32)      # bio_utils.py:214-220
            g[f"{name}/cmap_bits/{k}"] = np.packbits(cm.astype(np.uint8), axis=1)
            g[f"{name}/cmap_nnz/{k}"] = np.int64(cm.sum())
        seq = "A" * X.shape[0]
        cm6 = (D < 6.0 ** 2).astype(np.int32)
        sparse = np.argwhere(cm6 == 1).astype(np.int32)
        for gen in (0, 2):
            g[f"{name}/aligned_bits/gen{gen}"] = np.packbits(ref.align_contact_map(seq, seq, sparse, gen).astype(np.uint8), axis=1)
        sub = (D > 0) & (D < np.finfo(np.float32).tiny)
        print(f"{name}: subnormal d^2 cells {int(sub.sum())}, inf cells {int(np.isinf(D).sum())}, NaN cells {int(np.isnan(D).sum())}, "
              f"contacts per threshold {[int(g[f'{name}/cmap_nnz/{k}']) for k in range(len(THRESHOLDS))]}")
    g["numpy_version"] = np.array(np.__version__)
    path = os.path.join(HERE, "cmap_special_golden.npz")
    np.savez_compressed(path, **g)
    print(os.path.basename(path), len(g), "arrays,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
