#!/usr/bin/env python3
"""Golden vectors of the coords -> contact map -> aligned map chain at OTHER thresholds / generated_contacts than the CLI default
(6 A, 2), produced by the compiled reference kernels (oracle/_ref, built from /root/reference/mDeepFRI/contact_map_utils.pyx): the
released GCN files are `..._ca_10.0_...` models (reference mDeepFRI/__init__.py:73,78), and generated_contacts is a CLI option
(cli.py:360-371).  Writes tests/golden/cmap_thr_golden.npz: inputs + sha256 / packed bits of the reference's outputs (data only).

    python tests/golden/make_thr_golden.py        # needs /root/reference (build container only)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
sys.path.insert(0, ROOT)   # mdfri_testkit

import build_ref  # noqa: E402
from mdfri_testkit import synthetic


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = build_ref.load()
    if ref is None:
        build_ref.build()
        ref = build_ref.load()
    assert ref is not None, "reference build unavailable"
    g, names = {}, []
    for thr, gen in ((10.0, 2), (4.0, 0), (8.0, 5), (10.0, 0), (7.5, 1)):
        for L, rate in ((40, 0.1), (200, 0.05), (330, 0.12), (512, 0.0)):
            rng = np.random.default_rng(int(thr * 10) * 1000 + gen * 100 + L)
            seq = synthetic.random_sequence(rng, L)
            q, t, lt = synthetic.mutate_alignment(rng, seq, rate) if rate > 0 else (seq, seq, L)
            coords = synthetic.random_walk_coords(rng, lt)
            D = ref.pairwise_sqeuclidean(coords)                                  # bio_utils.py:196-227, mode="sparse"
            sparse = np.argwhere((D < thr**2).astype(np.int32) == 1).astype(np.int32)
            out = ref.align_contact_map(q, t, sparse, gen)                        # bio_utils.py:348-385
            name = f"thr/t{thr:g}_g{gen}_L{L}"
            g[f"{name}/coords"] = coords
            g[f"{name}/q"] = np.frombuffer(q.encode(), dtype=np.uint8)
            g[f"{name}/t"] = np.frombuffer(t.encode(), dtype=np.uint8)
            g[f"{name}/thr"] = np.float64(thr)
            g[f"{name}/gen"] = np.int32(gen)
            g[f"{name}/nnz_target"] = np.int64(sparse.shape[0])
            g[f"{name}/sha_sparse"] = np.frombuffer(sha(sparse).encode(), dtype=np.uint8)
            g[f"{name}/sha_out"] = np.frombuffer(sha(out).encode(), dtype=np.uint8)
            if L <= 200:
                g[f"{name}/out_bits"] = np.packbits(out.astype(np.uint8), axis=1)
            names.append(name)
    g["index"] = np.array(names)
    path = os.path.join(HERE, "cmap_thr_golden.npz")
    np.savez_compressed(path, **g)
    print(path, len(names), "cases,", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
