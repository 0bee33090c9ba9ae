"""Developer fuzz (run by hand from the repository root: python tests/fuzz/<name>.py; a seeded slice of it is in tests/test_gpu_gcn.py): random CNN topologies and random LM-model shapes / batch splits against the oracles."""
import os, sys
sys.path.insert(0, "metagenomic-deepfri_amd"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
import numpy as np
import cnn_oracle, lm_oracle, gcn_oracle
from mdfri_testkit import synthetic
from mDeepFRI.batch import SequenceEngine, HotPathEngine, PackedProteins
from mDeepFRI.predict import Predictor
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", 123)))
worst = 0
for it in range(25):
    nb = int(rng.integers(1, 6))
    ks = tuple(int(x) for x in rng.integers(1, 41, size=nb))
    fs = tuple(int(x) for x in rng.integers(1, 301, size=nb))
    w = synthetic.glorot_cnn_weights(seed=it, n_terms=int(rng.integers(1, 300)), filters=fs, kernel_lens=ks)
    if it % 3 == 0:
        for b, k in enumerate(ks, 1):
            w[f"cnn_pad{b}"] = np.array([int(rng.integers(0, k))], np.float32)
    seqs = [synthetic.random_sequence(rng, int(L)) for L in rng.integers(1, 400, size=int(rng.integers(1, 12)))]
    eng = SequenceEngine({"m": Predictor("c", weights=w)}, max_rows=int(rng.choice([128, 512, 1 << 20])))
    out = eng.run(seqs)["m"]
    for i, s in enumerate(seqs):
        e = float(np.abs(out[i] - cnn_oracle.cnn_forward(w, s)).max())
        worst = max(worst, e)
        assert e < 1e-4, (it, ks, fs, len(s), e)
print("cnn fuzz ok, worst", worst)
worst = 0
for it in range(8):
    H = int(rng.choice([64, 128]))
    E = int(rng.choice([256, 512]))
    w = synthetic.glorot_gcn_weights(seed=it, n_terms=int(rng.integers(1, 100)), embed=E, gc_dims=tuple(int(x) for x in rng.choice([256, 512], size=int(rng.integers(1, 4)))), fc_dim=256)
    w.update(synthetic.glorot_lm_weights(seed=100 + it, hidden=H, embed=E))
    n = int(rng.integers(1, 30))
    prots = synthetic.synthetic_proteins(seed=it, count=n, length=(1, 150), indel_rate=0.05)
    for form in ("0", None):
        if form is None: os.environ.pop("MDFRI_LM_PERSISTENT_MAX_B", None)
        else: os.environ["MDFRI_LM_PERSISTENT_MAX_B"] = form
        eng = HotPathEngine({"m": Predictor("g", weights=w)}, max_rows=int(rng.choice([128, 1024, 65536])), lm_batch=int(rng.choice([3, 8192])))
        pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots], max_rows=eng.max_rows)
        out = eng.run_alignments(pk)["m"]
        import cmap_oracle
        for i, p in enumerate(prots):
            A = cmap_oracle.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)
            e = float(np.abs(out[i] - lm_oracle.gcn_lm_forward(w, p["seq"], A)).max())
            worst = max(worst, e)
            assert e < 1e-4, (it, form, i, len(p["seq"]), e)
print("lm fuzz ok, worst", worst)
