"""Latency of the per-call drop-in API (what an unmodified pipeline.py loop pays per protein)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # mdfri_testkit (synthetic workloads)
from mDeepFRI import bio_utils
from mdfri_testkit import synthetic
from mDeepFRI.contact_map_utils import pairwise_sqeuclidean, align_contact_map
from mDeepFRI.predict import Predictor, seq2onehot


def timeit(fn, n=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


class Aln:
    pass


w = synthetic.glorot_gcn_weights(seed=0, n_terms=489)
pred = Predictor("syn", weights=w)
cnn = Predictor("syn-cnn", weights=synthetic.glorot_cnn_weights(seed=0, n_terms=489))
for L in (128, 512, 1024):
    p = synthetic.synthetic_proteins(seed=L, count=1, length=L, indel_rate=0.05)[0]
    a = Aln()
    a.coords, a.gapped_sequence, a.gapped_target, a.target_name, a.query_name = p["coords"], p["q_aln"], p["t_aln"], "t", "q"
    D = pairwise_sqeuclidean(p["coords"])
    sparse = np.argwhere((D < 36).astype(np.int32) == 1).astype(np.int32)
    _, cm = bio_utils.build_align_contact_map(a, 6.0, 2)
    print(f"L={L:5d}  pairwise_sqeuclidean {timeit(lambda: pairwise_sqeuclidean(p['coords'])):7.3f} ms   "
          f"align_contact_map {timeit(lambda: align_contact_map(p['q_aln'], p['t_aln'], sparse, 2)):7.3f} ms   "
          f"build_align_contact_map {timeit(lambda: bio_utils.build_align_contact_map(a, 6.0, 2)):7.3f} ms   "
          f"seq2onehot {timeit(lambda: seq2onehot(p['seq'])):6.3f} ms   "
          f"forward_pass(GCN) {timeit(lambda: pred.forward_pass(p['seq'], cm)):7.3f} ms   "
          f"forward_pass(CNN) {timeit(lambda: cnn.forward_pass(p['seq'])):7.3f} ms")
