"""Throughput of the batched Needleman-Wunsch aligner (csrc/nw.hip): score mode over candidate sets, then full alignments of
the winners -- host lists in, arrays out (mDeepFRI.alignment.align_queries_arrays), plus the bare kernels through the dev API."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))   # mdfri_testkit (synthetic workloads)
import torch  # noqa: E402
from mDeepFRI import _hip  # noqa: E402
from mdfri_testkit import synthetic
from mDeepFRI.alignment import ScoringMatrix, _PairBatch, align_queries_arrays  # noqa: E402

ALPHA = "ARNDCQEGHILKMFPSTWYVBZX*"
rng = np.random.default_rng(1)
m = rng.integers(-6, 4, size=(24, 24))
m = (m + m.T) // 2
np.fill_diagonal(m, rng.integers(5, 13, size=24))
sm = ScoringMatrix(ALPHA, m)
Q, K = int(os.environ.get("NWQ", 4000)), 8
lens = synthetic.histogram_lengths(3, 3000)
if os.environ.get("NW_UNIFORM"):      # all sequences the same length: throughput without the long-pair tail
    lens = np.full(3000, int(os.environ["NW_UNIFORM"]))
db = {f"t{k}": synthetic.random_sequence(rng, int(L)) for k, L in enumerate(lens)}
keys = list(db)
qseqs, cands = [], []
for i in range(Q):
    ks = [keys[j] for j in rng.integers(0, len(keys), size=K)]
    qseqs.append(db[ks[0]][: max(30, len(db[ks[0]]) - int(rng.integers(0, 20)))])
    cands.append({k: db[k] for k in ks})
align_queries_arrays(["w"] * Q, qseqs, cands, scoring_matrix=sm)   # warm-up (sizes the library's device scratch)
runs = []
for _ in range(5):
    t0 = time.perf_counter()
    batch = align_queries_arrays([f"q{i}" for i in range(Q)], qseqs, cands, scoring_matrix=sm)
    runs.append(time.perf_counter() - t0)
dt = sorted(runs)[2]
print("host-to-host runs (ms):", " ".join(f"{r * 1e3:.1f}" for r in runs))
cells_score = sum(len(q) * sum(len(v) for v in c.values()) for q, c in zip(qseqs, cands))
cells_full = sum(len(q) * len(t) for q, t in zip(batch.query_sequences, batch.target_sequences))
print(f"align_queries_arrays: {Q} queries x {K} candidates, mean L {np.mean([len(q) for q in qseqs]):.0f}: {dt * 1e3:.1f} ms host-to-host "
      f"= {Q / dt:.0f} queries/s, {(cells_score + cells_full) / dt / 1e9:.1f} GCUPS incl. encoding + copies")

# bare kernels: device-resident inputs, HIP events
L = _hip.lib()
seqs = list(qseqs)
index, pq, pt = {}, [], []
for qi, c in enumerate(cands):
    for k, t in c.items():
        if k not in index:
            index[k] = len(seqs)
            seqs.append(t)
        pq.append(qi)
        pt.append(index[k])
pb = _PairBatch(seqs, sm)
pq, pt = np.array(pq, np.int32), np.array(pt, np.int32)
if os.environ.get("NW_SORT", "1") == "1":      # largest DP matrices first (what mDeepFRI.alignment does)
    o = pb._by_cost(pq, pt)
    pq, pt = np.ascontiguousarray(pq[o]), np.ascontiguousarray(pt[o])
P = len(pq)
if os.environ.get("NW_ORIENT", "1") == "1":    # score mode: rows = whichever sequence takes fewer steps (what the host entries do)
    print("pairs turned round:", L.mdf_nw_orient_pairs(_hip.ptr(pb.seq_len), _hip.ptr(pq), _hip.ptr(pt), P, _hip.ptr(sm.matrix), 24, 10, 1), "of", P)
bnd_off = np.zeros(P + 1, np.int64)
L.mdf_nw_plan(_hip.ptr(pb.seq_len), _hip.ptr(pq), _hip.ptr(pt), P, _hip.ptr(bnd_off), None, None)
dev = torch.device("cuda:0")
up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
d = dict(codes=up(pb.codes), off=up(pb.seq_off), ln=up(pb.seq_len), pq=up(pq), pt=up(pt), mat=up(sm.matrix), bo=up(bnd_off))
bnd = torch.empty(int(bnd_off[-1]) + 1, dtype=torch.int32, device=dev)
sc = torch.empty(P, dtype=torch.int32, device=dev)
st = _hip.ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
n_long = int(L.mdf_nw_count_long(_hip.ptr(pb.seq_len), _hip.ptr(pq), _hip.ptr(pt), P)) if os.environ.get("NW_COOP", "1") == "1" else 0
run = lambda: _hip.check(L.mdf_nw_score_dev(_hip.ptr(d["codes"]), _hip.ptr(d["off"]), _hip.ptr(d["ln"]), _hip.ptr(d["pq"]), _hip.ptr(d["pt"]), P, n_long,  # noqa: E731
                                            _hip.ptr(d["mat"]), 24, 10, 1, _hip.ptr(d["bo"]), _hip.ptr(bnd), _hip.ptr(sc), st))
run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"k_nw<score> ({n_long} pairs swept by a workgroup each): {P} pairs, {cells_score / 1e9:.2f} G cells: {ms:.2f} ms = {cells_score / ms / 1e6:.0f} GCUPS, {P / ms * 1e3:.0f} pairs/s")

# where the host-to-host time goes
import cProfile  # noqa: E402
import pstats  # noqa: E402
pr = cProfile.Profile()
pr.enable()
align_queries_arrays([f"q{i}" for i in range(Q)], qseqs, cands, scoring_matrix=sm)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
