"""Small-batch rate of the fused path (B proteins of L=512, MF+BP+CC): proteins/s when one batch is one forward call -- the
serving shape, where the launch sequence (about 40 kernels) and not the kernels bounds the rate.  Prints, per B, the
launch-by-launch rate and the hipGraph-replay rate of the C++ engine.  `--pkg DIR` runs the same measurement against another
checkout of the package (e.g. the round-2 tree, whose chunk loop was Python) for the before/after table in DESIGN.md."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--pkg", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--sizes", default="1,8,64,512")
args = ap.parse_args()
sys.path.insert(0, os.path.abspath(args.pkg))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # mdfri_testkit (synthetic workloads)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from mdfri_testkit import synthetic
from mDeepFRI.batch import HotPathEngine, PackedProteins  # noqa: E402
from mDeepFRI.predict import Predictor  # noqa: E402

MODES = ("mf", "bp", "cc")
preds = {m: Predictor(m, weights=synthetic.glorot_gcn_weights(seed=i, n_terms=synthetic.GO_TERMS[m])) for i, m in enumerate(MODES)}
has_graph = "graph_max_chunks" in HotPathEngine.__init__.__code__.co_varnames
res = {"pkg": os.path.abspath(args.pkg), "engine_in_library": has_graph, "sizes": {}}
for B in [int(x) for x in args.sizes.split(",")]:
    prots = synthetic.synthetic_proteins(seed=B, count=B, length=512)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots], max_rows=65536)
    row = {}
    for mode in (("eager", "graph") if has_graph else ("python_loop",)):
        kw = {"graph_max_chunks": -1} if mode == "eager" else {}
        eng = HotPathEngine(preds, device=0, max_rows=65536, **kw)
        db = eng.upload(pk)
        out = eng.outputs_for(db) if has_graph else None
        step = (lambda: eng.forward_alignments(db, out=out)) if has_graph else (lambda: eng.forward_alignments(db))
        for _ in range(5):
            step()
        eng.check(db)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps
        row[mode] = {"us_per_call": round(dt * 1e6, 1), "proteins_per_s": round(B / dt, 1)}
    res["sizes"][str(B)] = row
print(json.dumps(res))
