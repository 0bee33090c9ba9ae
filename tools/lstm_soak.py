"""Soak test of the persistent LSTM kernel's device-wide barrier / hand-off (MI355X_MICROARCH.md: test every hand-off under uneven
load, checking every word): the recurrence is deterministic, so every repetition must reproduce the first one bit for bit -- idle,
and while another stream streams 1 GiB copies through the memory system."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # mdfri_testkit (synthetic workloads)
import torch
from mdfri_testkit import synthetic
from mDeepFRI.batch import HotPathEngine, PackedProteins
from mDeepFRI.predict import Predictor

w = synthetic.glorot_gcn_weights(seed=0, n_terms=64)
w.update(synthetic.glorot_lm_weights(seed=1000))
eng = HotPathEngine({"mf": Predictor("syn", weights=w)}, max_rows=65536)
rng = np.random.default_rng(0)
side = torch.cuda.Stream()
a = torch.empty(256 << 20, dtype=torch.float32, device="cuda")
b = torch.empty_like(a)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
for B, L in ((1, 512), (16, 300), (64, 200), (300, 64)):
    seqs = [synthetic.random_sequence(rng, int(x)) for x in rng.integers(max(1, L // 2), L + 1, size=B)]
    pk = PackedProteins.pack(seqs, max_rows=65536)
    ref = eng.lm_features(pk)
    bad = 0
    t0 = time.perf_counter()
    for it in range(reps):
        if it % 2:   # every other repetition under memory load from another stream
            with torch.cuda.stream(side):
                for _ in range(4):
                    b.copy_(a, non_blocking=True)
        got = eng.lm_features(pk)
        for x, y in zip(ref, got):
            if not np.array_equal(x, y):
                bad += 1
                break
    torch.cuda.synchronize()
    print(f"B={B:4d} Lmax={L:4d}: {reps} repetitions, {bad} mismatching, {time.perf_counter() - t0:.1f} s")
    assert bad == 0
print("soak ok")
