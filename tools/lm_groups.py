"""LSTM features for groups of B proteins (L=512): persistent one-launch form vs per-step GEMM form."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # mdfri_testkit (synthetic workloads)
import torch
from mDeepFRI import _hip
from mdfri_testkit import synthetic
lib = _hip.lib()
from mDeepFRI.batch import HotPathEngine, PackedProteins
from mDeepFRI.predict import Predictor

w = synthetic.glorot_gcn_weights(seed=0, n_terms=64)
w.update(synthetic.glorot_lm_weights(seed=1000))
pred = Predictor("syn", weights=w)
eng = HotPathEngine({"mf": pred}, max_rows=65536)
rng = np.random.default_rng(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for B in (1, 4, 16, 64, 128, 256, 512, 1024):
    seqs = [synthetic.random_sequence(rng, L) for _ in range(B)]
    pk = PackedProteins.pack(seqs, max_rows=65536)
    res = {}
    for form, knob in (("persistent", "1024"), ("gemm", "0")):
        os.environ["MDFRI_LM_PERSISTENT_MAX_B"] = knob
        f = eng.lm_features(pk)
        torch.cuda.synchronize()
        lib.mdf_timing_reset(); lib.mdf_timing_enable(1)
        f = eng.lm_features(pk)
        torch.cuda.synchronize()
        lib.mdf_timing_enable(0)
        n, ms = _hip.c_int64(0), _hip.ctypes.c_double(0.0)
        tot = 0.0
        for kind in (b"lstm", b"lstm2"):
            lib.mdf_timing_read(kind, n, ms); tot = max(tot, ms.value)
        res[form] = (tot * 1e-3, f)     # GPU time of the recurrence only (events), not the host copy of the features
    d = max(float(np.abs(a - b).max()) for a, b in zip(res["persistent"][1], res["gemm"][1]))
    print(f"B={B:5d}  persistent {res['persistent'][0]*1e3:8.1f} ms ({B/res['persistent'][0]:8.0f}/s)   gemm {res['gemm'][0]*1e3:8.1f} ms ({B/res['gemm'][0]:8.0f}/s)   max diff {d:.1e}")
