import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("metagenomic-deepfri_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
from mdfri_testkit import synthetic
from mDeepFRI.batch import HotPathEngine, PackedProteins
from mDeepFRI.predict import Predictor
w = synthetic.glorot_gcn_weights(seed=0, n_terms=64)
pred = Predictor("syn", weights=w)
thr, gen, nnz = float(os.environ.get("THR", 6.0)), int(os.environ.get("GEN", 2)), int(os.environ.get("NNZ", 24))
prots = synthetic.synthetic_proteins(seed=102, count=18, length=(20, 330), indel_rate=float(os.environ.get("INDEL", 0.12)))
eng = HotPathEngine({"mf": pred}, device=0, max_rows=1024, nnz_per_row=nnz, threshold=thr, generated_contacts=gen)
pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots], max_rows=1024)
out = eng.run_alignments(pk)["mf"]
np.save(sys.argv[1], out)
print([len(p["seq"]) for p in prots], [(c.first, c.count, c.rows) if hasattr(c, "first") else c.rows for c in pk.chunks])
