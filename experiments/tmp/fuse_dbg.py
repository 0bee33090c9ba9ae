import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ("metagenomic-deepfri_amd", "oracle", ""):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
from mdfri_testkit import synthetic
from mDeepFRI.batch import HotPathEngine, PackedProteins
from mDeepFRI.predict import Predictor
w = synthetic.glorot_gcn_weights(seed=5, n_terms=96)
pred = Predictor("syn", weights=w)
prots = synthetic.synthetic_proteins(seed=71, count=12, length=(180, 250)) + synthetic.synthetic_proteins(seed=72, count=6, length=(400, 512))
pk = PackedProteins.pack([q["seq"] for q in prots], [q["coords"] for q in prots], [q["q_aln"] for q in prots], [q["t_aln"] for q in prots], max_rows=int(os.environ.get("MR", 32768)))
eng = HotPathEngine({"a": pred}, device=0, max_rows=int(os.environ.get("MR", 32768)))
out, logits = eng.forward_alignments(eng.upload(pk), want_logits=True)
np.save(sys.argv[1], logits["a"].cpu().numpy())
