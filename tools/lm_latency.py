"""Per-call latency of Predictor.forward_pass on a model with the language-model branch (L=512), persistent vs GEMM LSTM form."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # mdfri_testkit (synthetic workloads)
from mdfri_testkit import synthetic
from mDeepFRI.predict import Predictor

w = synthetic.glorot_gcn_weights(seed=0, n_terms=489)
w.update(synthetic.glorot_lm_weights(seed=1000))
pred = Predictor("syn", weights=w)
rng = np.random.default_rng(0)
for L in (128, 512, 1000):
    seq = synthetic.random_sequence(rng, L)
    xyz = synthetic.random_walk_coords(rng, L)
    A = (((xyz[:, None] - xyz[None]) ** 2).sum(-1) < 36).astype(np.int32)
    for form in ("persistent", "gemm"):
        if form == "gemm":
            os.environ["MDFRI_LM_PERSISTENT_MAX_B"] = "0"
        else:
            os.environ.pop("MDFRI_LM_PERSISTENT_MAX_B", None)
        y = pred.forward_pass(seq, A)
        t0 = time.perf_counter()
        for _ in range(5):
            y2 = pred.forward_pass(seq, A)
        dt = (time.perf_counter() - t0) / 5
        print(f"L={L:5d} {form:10s} {dt*1e3:8.2f} ms/call   max|dy| vs first form {0.0 if form=='persistent' else float(np.abs(y2-yp).max()):.2e}")
        if form == "persistent":
            yp = y2
