"""A.X kernel probe: mean launch time of the A.X kernel (HIP events around every launch) on four workload shapes plus a sha256 of
all scores, so that two builds of the library (MDFRI_HIP_LIB=...) can be compared bit for bit.  The dense random maps at the end go
through forward_pass (L/2 entries per row)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # mdfri_testkit (synthetic workloads)
import torch  # noqa: E402
from mDeepFRI import _hip  # noqa: E402
from mdfri_testkit import synthetic
from mDeepFRI.batch import HotPathEngine, PackedProteins  # noqa: E402
from mDeepFRI.predict import Predictor  # noqa: E402

lib = _hip.lib()
w = synthetic.glorot_gcn_weights(seed=0, n_terms=489)
pred = Predictor("probe", weights=w)
eng = HotPathEngine({"mf": pred}, max_rows=65536)
h = hashlib.sha256()
for name, prots in (("L512 x 512", synthetic.synthetic_proteins(3, 512, 512)),
                    ("L256 x 1024", synthetic.synthetic_proteins(4, 1024, 256)),
                    ("L1024 x 256", synthetic.synthetic_proteins(5, 256, 1024)),
                    ("mixed 128..1024 x 448, 5% indels", sorted(synthetic.synthetic_proteins(6, 448, (128, 1024), 0.05), key=lambda p: len(p["seq"])))):
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots], max_rows=65536)
    db = eng.upload(pk)
    for _ in range(2):
        out = eng.forward_alignments(db)
    eng.check(db)
    lib.mdf_timing_reset()
    lib.mdf_timing_enable(1)
    for _ in range(5):
        out = eng.forward_alignments(db)
    eng.check(db)
    lib.mdf_timing_enable(0)
    n, ms = _hip.c_int64(0), _hip.ctypes.c_double(0.0)
    lib.mdf_timing_read(b"ax", n, ms)
    rows = sum(c.rows for c in pk.chunks) / len(pk.chunks)
    nnz = float(eng.last_chunk_nnz()) / pk.chunks[-1].rows
    us = 1e3 * ms.value / max(n.value, 1)
    gbs = rows * (2 * 4 * 512 + 4 + 8 * nnz) / (us * 1e-6) / 1e9
    h.update(out["mf"].cpu().numpy().tobytes())
    print(f"{name:36s} rows/launch {rows:8.0f}  nnz/row {nnz:5.2f}  A.X {us:7.2f} us/launch  {gbs:7.1f} GB/s  frac {gbs / 8000:.3f}", flush=True)
# dense random 0/1 map (the reference notebook's recipe): L/2 entries per row -> direct-gather branch
rng = np.random.default_rng(9)
for L in (96, 700):
    seq = synthetic.random_sequence(rng, L)
    A = rng.integers(0, 2, size=(L, L)).astype(np.int32)
    h.update(pred.forward_pass(seq, A).tobytes())
print("library", os.environ.get("MDFRI_HIP_LIB", "default"), "sha256", h.hexdigest(), flush=True)
