"""Repeat each batched path several times on the same input: results must be bit-identical (no float atomics anywhere; the CNN's
integer atomicMax is order-independent; the two-stream LSTM form is ordered by events)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # mdfri_testkit (synthetic workloads)
from mdfri_testkit import synthetic
from mDeepFRI.batch import HotPathEngine, PackedProteins, SequenceEngine
from mDeepFRI.predict import Predictor

prots = synthetic.synthetic_proteins(seed=5, count=700, length=(50, 600), indel_rate=0.05)
seqs = [p["seq"] for p in prots]
pk = PackedProteins.pack(seqs, [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots], max_rows=65536)
w = synthetic.glorot_gcn_weights(seed=0, n_terms=489)
wl = dict(w); wl.update(synthetic.glorot_lm_weights(seed=1000))
for name, eng, run in (
        ("gcn", HotPathEngine({"m": Predictor("g", weights=w)}, max_rows=65536), lambda e: e.run_alignments(pk)["m"]),
        ("gcn+lm (GEMM-form LSTM, 700 proteins)", HotPathEngine({"m": Predictor("l", weights=wl)}, max_rows=65536), lambda e: e.run_alignments(pk)["m"]),
        ("cnn", SequenceEngine({"m": Predictor("c", weights=synthetic.glorot_cnn_weights(seed=1, n_terms=489))}), lambda e: e.run(seqs)["m"])):
    ref = run(eng)
    for _ in range(8):
        assert np.array_equal(ref, run(eng)), name
    print(name, "bit-identical over 9 runs", ref.shape)
