"""Rate of the batch entry points on a SHUFFLED mixed-length batch: the plan that visits the proteins as given (keep_order), the default plan (visits
them shortest first, results in input order) and the same proteins handed over already sorted -- VERDICT r4 #5: within 2 % of the sorted rate, and the
same bits.  Run on an MI355X: python tools/sorted_plan_rate.py [proteins]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("metagenomic-deepfri_amd", ""):
    sys.path.insert(0, os.path.join(ROOT, d))
import torch  # noqa: E402
from mdfri_testkit import synthetic  # noqa: E402
from mDeepFRI import _hip  # noqa: E402
from mDeepFRI.batch import HotPathEngine, PackedProteins  # noqa: E402
from mDeepFRI.predict import Predictor  # noqa: E402

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    lengths = synthetic.uniform_lengths(46, n)
    seqs, coords, qa, ta = synthetic.bulk_proteins(46, lengths, range(n), indel_rate=0.05, workers=0)      # (serial: every protein has its own seeded generator)
    preds = {m: Predictor(f"syn-{m}", weights=synthetic.glorot_gcn_weights(seed=k, n_terms=synthetic.GO_TERMS[m], sparse_scores=True)) for k, m in enumerate(("mf", "bp", "cc"))}
    eng = HotPathEngine(preds, device=0, max_rows=65536)
    cols = lambda idx: ([seqs[i] for i in idx], [coords[i] for i in idx], [qa[i] for i in idx], [ta[i] for i in idx])  # noqa: E731


    def rate(pk, reps=3):
        db = eng.upload(pk)
        out = eng.forward_alignments(db)
        eng.check(db)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = eng.forward_alignments(db)
        torch.cuda.synchronize()
        return n * reps / (time.perf_counter() - t0), {m: t.cpu().numpy() for m, t in out.items()}


    order = np.argsort(lengths, kind="stable")
    r_keep, o_keep = rate(PackedProteins.pack(seqs, coords, qa, ta, max_rows=65536, keep_order=True))
    r_plan, o_plan = rate(PackedProteins.pack(seqs, coords, qa, ta, max_rows=65536))
    r_sorted, o_sorted = rate(PackedProteins.pack(*cols(order), max_rows=65536))
    same = all(np.array_equal(o_plan[m], o_keep[m]) and np.array_equal(o_plan[m][order], o_sorted[m]) for m in preds)
    print(f"{n} mixed-length proteins (L ~ U[128, 1024], shuffled), three heads, device-resident inputs, proteins/s:")
    print(f"  visited as given (keep_order)      {r_keep:10.0f}")
    print(f"  default plan (sorts inside)        {r_plan:10.0f}   = {r_plan / r_sorted:.3f} x the pre-sorted rate")
    print(f"  handed over sorted                 {r_sorted:10.0f}")
    print(f"  bit-identical across the three: {same}")
    # the single-call C entry (host arrays in, host arrays out) on the shuffled batch
    sb, qb, tb = "".join(seqs).encode(), "".join(qa).encode(), "".join(ta).encode()
    Lq = np.array([len(s) for s in seqs], dtype=np.int32)
    Lt = np.array([c.shape[0] for c in coords], dtype=np.int32)
    La = np.array([len(s) for s in qa], dtype=np.int32)
    xyz = np.ascontiguousarray(np.concatenate(coords, axis=0), dtype=np.float32)
    outs = [np.empty((n, preds[m].n_terms), dtype=np.float32) for m in eng.modes]
    ptrs = (ctypes.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
    for rep in range(2):
        t0 = time.perf_counter()
        rc = eng.L.mdf_engine_run_alignments_host(eng.handle, sb, _hip.ptr(Lq), n, _hip.ptr(xyz), _hip.ptr(Lt), qb, tb, _hip.ptr(La), ptrs, None)
        dt = time.perf_counter() - t0
    assert rc == 0, _hip.last_error()
    print(f"  mdf_engine_run_alignments_host     {n / dt:10.0f}   (host arrays -> host arrays, upload + download inside); identical: "
          f"{all(np.array_equal(o, o_keep[m]) for m, o in zip(eng.modes, outs))}")


if __name__ == "__main__":
    main()
