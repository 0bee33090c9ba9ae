"""The widened path end to end (tools/pipeline_example.py: GPU aligner -> packed arrays -> fused contact map + GCN -> GPU filter)
against the oracle chain stage by stage: nw_oracle (best hit, operations) -> cmap_oracle -> gcn_oracle -> output_oracle."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_pipeline_example_matches_the_oracle_chain():
    import cmap_oracle
    import gcn_oracle
    import nw_oracle
    from mDeepFRI.alignment import insert_gaps
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pipeline_example
    batch, scores, kept, (qids, qseqs, cands, db_xyz, weights, sm) = pipeline_example.main(160)
    assert kept == list(range(160))
    s_mf = scores["mf"].cpu().numpy()
    for i in range(0, 160, 13):
        key, tseq = nw_oracle.best_hit_database(qseqs[i], cands[i], sm.matrix, sm.alphabet)
        ops = nw_oracle.align_pairwise(qseqs[i], tseq, sm.matrix, sm.alphabet)[0]
        assert batch.target_keys[i] == key
        a, b = int(batch.aln_off[i]), int(batch.aln_off[i + 1])
        assert bytes(batch.ops[a:b]).decode() == ops
        gq, gt = insert_gaps(qseqs[i], tseq, ops)
        cm = cmap_oracle.build_align_contact_map(db_xyz[key], gq, gt, 6.0, 2)
        assert np.max(np.abs(s_mf[i] - gcn_oracle.gcn_forward(weights["mf"], qseqs[i], cm))) < 1e-4, i
