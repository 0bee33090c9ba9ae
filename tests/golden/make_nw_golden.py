"""Regression anchors for the alignment step: outputs of oracle/nw_oracle.c (PARITY UNPINNED against PyOpal, see its header) on
seeded inputs, plus the reference's own known answers (mDeepFRI/tests/test_alignment.py:9-48).  Run: python tests/golden/make_nw_golden.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
import nw_oracle as nwo  # noqa: E402

ALPHA = "ARNDCQEGHILKMFPSTWYVBZX*"
rng = np.random.default_rng(2024)
M = rng.integers(-6, 4, size=(24, 24))
M = (M + M.T) // 2
np.fill_diagonal(M, rng.integers(5, 13, size=24))
out = {"alphabet": np.frombuffer(ALPHA.encode(), np.uint8), "matrix": M.astype(np.int32)}
names = []


def seq(n):
    return "".join(rng.choice(list(ALPHA[:20]), size=n))


def mutate(s, rate):
    o = []
    for c in s:
        r = rng.random()
        if r < rate / 3:
            continue
        if r < 2 * rate / 3:
            o.append(rng.choice(list(ALPHA[:20])))
        o.append(c if rng.random() > rate else rng.choice(list(ALPHA[:20])))
    return "".join(o) or "A"


cases = [("kat", "MAGFLKVVQLLAKYGSKAVQWAWANKGKILDWLNAGQAIDWVVS", "MAGFLKVVQILAKYGSKAVQWAWANKGKILDWINAGQAIDWVVE", 10, 1)]
for k, (lq, rate) in enumerate([(1, 0.0), (5, 0.5), (63, 0.1), (64, 0.2), (65, 0.3), (130, 0.15), (300, 0.1), (520, 0.25)]):
    q = seq(lq)
    for go, ge in ((10, 1), (3, 2)):
        cases.append((f"r{k}_{go}_{ge}", q, mutate(q, rate), go, ge))
cases.append(("unrelated", seq(90), seq(140), 10, 1))
for name, q, t, go, ge in cases:
    ops, iden, _, _, score = nwo.align_pairwise(q, t, M, ALPHA, go, ge)
    names.append(name)
    out[f"{name}/q"] = np.frombuffer(q.encode(), np.uint8)
    out[f"{name}/t"] = np.frombuffer(t.encode(), np.uint8)
    out[f"{name}/gap"] = np.array([go, ge], np.int32)
    out[f"{name}/ops"] = np.frombuffer(ops.encode(), np.uint8)
    out[f"{name}/score"] = np.array(score, np.int32)
    out[f"{name}/identity"] = np.array(iden, np.float64)
out["index"] = np.array(names)
np.savez_compressed(os.path.join(HERE, "nw_golden.npz"), **out)
print(len(names), "cases ->", os.path.join(HERE, "nw_golden.npz"))
