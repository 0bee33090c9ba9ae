"""Run as a fresh process by tests/test_gpu_binding.py: the reference's actual call pattern for the contact-map stage --
`with Pool(threads) as p: p.map(partial(build_align_contact_map, threshold=..., generated_contacts=...), alignments)`
(reference pipeline.py:476-481) -- from a parent that has NOT touched the GPU, with forked workers that each initialise HIP
lazily on their first call.  Prints one JSON line."""
import json
import multiprocessing
import os
import sys
from functools import partial

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)   # mdfri_testkit

from mdfri_testkit import synthetic
from mDeepFRI.alignment import AlignmentResult  # noqa: E402
from mDeepFRI.bio_utils import build_align_contact_map  # noqa: E402


def main():
    prots = synthetic.synthetic_proteins(seed=91, count=24, length=(40, 300), indel_rate=0.06)
    alns = []
    for i, p in enumerate(prots):
        a = AlignmentResult(query_name=f"q{i}", query_sequence=p["seq"], target_name=f"t{i}", target_sequence=p["t_aln"].replace("-", ""),
                            coords=p["coords"] if i != 5 else None)      # one hit without coordinates -> (aln, None)
        a.gapped_sequence, a.gapped_target = p["q_aln"], p["t_aln"]
        alns.append(a)
    ctx = multiprocessing.get_context("fork")        # the reference uses the platform default (fork on Linux)
    with ctx.Pool(2) as pool:
        res = pool.map(partial(build_align_contact_map, threshold=6.0, generated_contacts=2), alns)
    import cmap_oracle
    ok, none = 0, 0
    for (a, cm), p in zip(res, prots):
        if cm is None:
            none += 1
            continue
        ok += int(np.array_equal(cm, cmap_oracle.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)) and cm.dtype == np.int32)
    print(json.dumps({"ok": ok, "none": none, "n": len(alns), "names": [a.query_name for a, _ in res[:3]]}))


if __name__ == "__main__":
    main()
