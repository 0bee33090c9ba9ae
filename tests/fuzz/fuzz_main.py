"""Developer fuzz (run by hand from the repository root: python tests/fuzz/<name>.py; a seeded slice of it is in tests/test_gpu_gcn.py): random GCN head shapes, chunk sizes, CSR capacities and workloads against the oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))   # mdfri_testkit
import numpy as np
import cmap_oracle, gcn_oracle
from mdfri_testkit import synthetic
from mDeepFRI.batch import HotPathEngine, PackedProteins
from mDeepFRI.predict import Predictor
SEED = int(os.environ.get("FUZZ_SEED", 7))
rng = np.random.default_rng(SEED)
worst = 0.0
for it in range(int(os.environ.get("FUZZ_ITERS", 16))):
    n_gc = int(rng.integers(1, 4))
    w = synthetic.glorot_gcn_weights(seed=SEED * 100 + it, n_terms=int(rng.integers(1, 2100)), embed=int(rng.choice([7, 64, 1024])),
                                     gc_dims=tuple(int(x) for x in rng.choice([256, 512, 1024], size=n_gc)), fc_dim=int(rng.choice([256, 512, 1024])))
    n = int(rng.integers(1, 40))
    prots = synthetic.synthetic_proteins(seed=1000 * SEED + it, count=n, length=(1, int(rng.choice([40, 300, 700]))), indel_rate=float(rng.choice([0.0, 0.05, 0.3])))
    thr, gen = float(rng.choice([4.0, 6.0, 10.0])), int(rng.integers(0, 5))
    eng = HotPathEngine({"m": Predictor("g", weights=w)}, max_rows=int(rng.choice([128, 2048, 65536])), nnz_per_row=int(rng.choice([2, 40])),
                        threshold=thr, generated_contacts=gen)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots],
                             max_rows=eng.max_rows)
    out = eng.run_alignments(pk)["m"]
    for i, p in enumerate(prots):
        A = cmap_oracle.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], thr, gen)
        e = float(np.abs(out[i] - gcn_oracle.gcn_forward(w, p["seq"], A)).max())
        worst = max(worst, e)
        assert e < 1e-4, (it, i, len(p["seq"]), e)
print("main-path fuzz ok, worst", worst)
