"""The widened path end to end on one GPU, the way reference pipeline.predict_protein_function strings it together
(pipeline.py:322-660), with every stage on the device:

    queries + candidate targets (sequences)  --align_queries_arrays-->  best hit + gapped strings     (alignment.py:164-320)
    + the targets' C-alpha traces            --PackedProteins.from_aligned_batch--> packed batch       (bio_utils.py:348-385)
    --HotPathEngine.forward_alignments-->  GO scores, 3 heads                                          (pipeline.py:292-319)
    --filter_scores / results_rows-->      results.tsv lines                                           (pipeline.py:684-748)

Synthetic data (no network): a database of random-walk structures, queries = mutated database members, 8 candidates each.
Prints the stage times and the overall proteins/s of ONE batch run stage after stage, then the sustained rate of the same
chain as a stream over several batches (mDeepFRI.stream.QueryStream: the host stages of batch k+1 run under the GCN of batch k)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
sys.path.insert(0, ROOT)   # mdfri_testkit (synthetic workloads)
import torch  # noqa: E402
from mdfri_testkit import synthetic
from mDeepFRI.alignment import ScoringMatrix, align_queries_arrays  # noqa: E402
from mDeepFRI.batch import HotPathEngine, PackedProteins  # noqa: E402
from mDeepFRI.output import filter_scores, results_text  # noqa: E402
from mDeepFRI.predict import Predictor  # noqa: E402

MODES = ("mf", "bp", "cc")
THRESHOLD = 0.1       # the reference's results.tsv filter (pipeline.py:696-705); the heads below are built sparse_scores=True: a few dozen terms per protein pass it


def make_inputs(n_queries, n_db, seed=0, k=8):
    rng = np.random.default_rng(seed)
    lens = synthetic.histogram_lengths(seed + 1, n_db)
    db_seq = {f"T{j}": synthetic.random_sequence(rng, int(L)) for j, L in enumerate(lens)}
    db_xyz = {name: synthetic.random_walk_coords(rng, len(s)) for name, s in db_seq.items()}
    names = list(db_seq)
    qids, qseqs, cands = [], [], []
    for i in range(n_queries):
        home = names[int(rng.integers(0, n_db))]
        q, _, _ = synthetic.mutate_alignment(rng, db_seq[home], 0.04)       # gapped query of a mutated copy ...
        qids.append(f"Q{i}")
        qseqs.append(q.replace("-", "") or "A")                              # ... ungapped: what a FASTA record holds
        others = [names[int(j)] for j in rng.integers(0, n_db, size=k - 1)]
        cands.append({n: db_seq[n] for n in [home] + others})
    return qids, qseqs, cands, db_xyz


def main(n_queries=int(os.environ.get("NQ", 4000))):
    sm = ScoringMatrix.simple()      # VTML80 needs the scoring_matrices package (absent offline); any matrix works the same way
    qids, qseqs, cands, db_xyz = make_inputs(n_queries, 1500)
    weights = {m: synthetic.glorot_gcn_weights(seed=i, n_terms=synthetic.GO_TERMS[m], sparse_scores=True) for i, m in enumerate(MODES)}
    eng = HotPathEngine({m: Predictor(f"syn-{m}", weights=weights[m]) for m in MODES}, max_rows=65536)
    terms = {m: [f"GO:{k:07d}" for k in range(synthetic.GO_TERMS[m])] for m in MODES}
    t = [time.perf_counter()]
    for rep in range(2):             # first pass warms allocations up
        t = [time.perf_counter()]
        batch = align_queries_arrays(qids, qseqs, cands, scoring_matrix=sm)
        t.append(time.perf_counter())
        packed, kept = PackedProteins.from_aligned_batch(batch, [db_xyz[k] for k in batch.target_keys], max_rows=65536)
        t.append(time.perf_counter())
        db = eng.upload(packed)
        scores = eng.forward_alignments(db)
        eng.check(db)
        t.append(time.perf_counter())
        n_lines = 0
        for m in MODES:
            off, ti, sc = filter_scores(scores[m], threshold=THRESHOLD, capacity_per_protein=synthetic.GO_TERMS[m])
            if rep and m == "cc":
                n_lines = results_text([qids[i] for i in kept], "gcn", m, terms[m], terms[m], off, ti, sc).count(b"\n")
        torch.cuda.synchronize()
        t.append(time.perf_counter())
    home_hit = float(np.mean([batch.target_keys[i] == list(cands[i])[0] for i in range(n_queries)]))
    d = np.diff(t)
    print(f"{n_queries} queries x 8 candidates, mean query length {np.mean([len(s) for s in qseqs]):.0f}; best hit = the mutated-from target for {100 * home_hit:.1f} % of the queries")
    print(f"  align (score 8 candidates + full alignment of the winner)   {d[0] * 1e3:8.1f} ms")
    print(f"  pack aligner arrays + C-alpha traces (no per-protein objects) {d[1] * 1e3:6.1f} ms")
    print(f"  upload + contact maps + GCN, 3 heads                        {d[2] * 1e3:8.1f} ms")
    print(f"  GPU filter (score >= {THRESHOLD}) of all heads + results.tsv lines of one head ({n_lines} lines) {d[3] * 1e3:6.1f} ms")
    print(f"  end to end {sum(d) * 1e3:.1f} ms = {n_queries / sum(d):.0f} proteins/s (host sequences in -> result lines out)")
    return batch, scores, kept, (qids, qseqs, cands, db_xyz, weights, sm)


def stream_main(n_batches=int(os.environ.get("NB", 6)), batch=int(os.environ.get("NQ", 4000))):
    """The same chain as a stream: n_batches x batch queries, result text of all three heads."""
    from mDeepFRI.stream import QueryStream
    sm = ScoringMatrix.simple()
    qids, qseqs, cands, db_xyz = make_inputs(n_batches * batch, 1500, seed=1)
    weights = {m: synthetic.glorot_gcn_weights(seed=i, n_terms=synthetic.GO_TERMS[m], sparse_scores=True) for i, m in enumerate(MODES)}
    eng = HotPathEngine({m: Predictor(f"syn-{m}", weights=weights[m]) for m in MODES}, max_rows=65536)
    terms = {m: [f"GO:{k:07d}" for k in range(synthetic.GO_TERMS[m])] for m in MODES}
    qs = QueryStream(eng, db_xyz, batch_size=batch, scoring_matrix=sm, threshold=THRESHOLD, batch_chunks=int(os.environ.get("BATCH_CHUNKS", 0)),
                     sort_by_length=os.environ.get("SORT", "1") != "0")
    for rep in range(2):             # first pass warms allocations up
        t0 = time.perf_counter()
        n_lines = n_bytes = 0
        for first, b, kept, res in qs.run(qids, qseqs, cands):
            for m in MODES:
                text = results_text([b.query_ids[i] for i in kept], "gcn", m, terms[m], terms[m], *res[m])
                n_lines += text.count(b"\n")
                n_bytes += len(text)
        dt = time.perf_counter() - t0
    print(f"stream of {n_batches} batches x {batch} queries (8 candidates each), three heads: {dt * 1e3:.1f} ms = {n_batches * batch / dt:.0f} proteins/s sustained "
          f"(host sequences in -> {n_lines} result lines, {n_bytes / 1e6:.1f} MB of text, out)")
    return n_batches * batch / dt


if __name__ == "__main__":
    main()
    stream_main()
