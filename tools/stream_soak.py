"""QueryStream soak: the filtered results of 12 000 queries must be bit-identical whatever the batch size and however often the stream runs
(the kernels are invariant to a protein's company; the software pipeline and its streams must not change that)."""
import os, sys, time, hashlib
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pipeline_example as pe
from mdfri_testkit import synthetic
from mDeepFRI.alignment import ScoringMatrix
from mDeepFRI.batch import HotPathEngine
from mDeepFRI.predict import Predictor
from mDeepFRI.stream import QueryStream
from mDeepFRI.output import results_text


def main():
    sm = ScoringMatrix.simple()
    qids, qseqs, cands, db_xyz = pe.make_inputs(12000, 1500, seed=7)
    w = {m: synthetic.glorot_gcn_weights(seed=i, n_terms=synthetic.GO_TERMS[m], sparse_scores=True) for i, m in enumerate(pe.MODES)}
    eng = HotPathEngine({m: Predictor(f"syn-{m}", weights=w[m]) for m in pe.MODES}, max_rows=65536)
    terms = {m: [f"GO:{k:07d}" for k in range(synthetic.GO_TERMS[m])] for m in pe.MODES}
    digests = []
    for rep, bs in enumerate((1000, 1000, 700, 3000, 1000)):
        qs = QueryStream(eng, db_xyz, batch_size=bs, scoring_matrix=sm)
        h = hashlib.sha256()
        t0 = time.perf_counter()
        per_query = {}
        for r in qs.run(qids, qseqs, cands):
            for m in pe.MODES:
                off, ti, sc = r.gcn[m]
                for k, i in enumerate(r.kept):
                    per_query[(r.first + r.aligned[i], m)] = (ti[off[k]:off[k + 1]].tobytes(), sc[off[k]:off[k + 1]].tobytes())
        for key in sorted(per_query):
            h.update(per_query[key][0]); h.update(per_query[key][1])
        digests.append(h.hexdigest())
        print(f"rep {rep} batch_size {bs}: {(time.perf_counter() - t0) * 1e3:.0f} ms, digest {digests[-1][:16]}")
    assert len(set(digests)) == 1, digests
    print("stream soak ok: identical results for every batch size and repetition")


if __name__ == "__main__":
    main()
