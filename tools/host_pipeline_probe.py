"""Round 6: where the host time of the two-slot host pipeline goes (mdf_engine_submit_alignments_host / mdf_engine_collect_host), through the
ctypes wrapper (mDeepFRI.batch.HostPipeline) and through the compiled binding (tests/binding/predict.pyx BatchEngine): per batch the
seconds spent inside submit (Python packing + the C call) and inside collect (wait + copy-out), next to the device's time per batch."""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from mDeepFRI import batch, weights as wfile  # noqa: E402
from mDeepFRI.predict import Predictor  # noqa: E402
from mdfri_testkit import synthetic  # noqa: E402

n, L, nb = int(os.environ.get("N", 10000)), 512, int(os.environ.get("BATCHES", 6))
w = {m: synthetic.glorot_gcn_weights(seed=i, n_terms=synthetic.GO_TERMS[m]) for i, m in enumerate(bench.MODES)}
two = [bench.make_fixed_length(777 + k, n, L) for k in range(2)]
batches = [two[k & 1] for k in range(nb)]


def drive(name, runner):
    runner.submit(*batches[0])
    runner.collect()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sub, col, done, inflight = [], [], [], 0
    for b in batches:
        t = time.perf_counter()
        runner.submit(*b)
        sub.append(time.perf_counter() - t)
        inflight += 1
        if inflight == 2:
            t = time.perf_counter()
            runner.collect()
            col.append(time.perf_counter() - t)
            done.append(time.perf_counter())
            inflight -= 1
    while inflight:
        t = time.perf_counter()
        runner.collect()
        col.append(time.perf_counter() - t)
        done.append(time.perf_counter())
        inflight -= 1
    print(f"{name:10s} submit ms {[round(1e3 * x, 1) for x in sub]}  collect ms {[round(1e3 * x, 1) for x in col]}  "
          f"steady {(nb - 1) * n / (done[-1] - done[0]):.0f} proteins/s", flush=True)


eng = batch.HotPathEngine({m: Predictor("p-" + m, weights=w[m]) for m in bench.MODES})
drive("ctypes", batch.HostPipeline(eng))
import binding_loader  # noqa: E402
_, bp = binding_loader.load()
with tempfile.TemporaryDirectory() as td:
    bpreds = []
    for m in bench.MODES:
        wfile.save_mdfw(os.path.join(td, m + ".mdfw"), w[m])
        bpreds.append(bp.Predictor(os.path.join(td, m + ".mdfw")))
drive("binding", bp.BatchEngine(bpreds))
drive("ctypes", batch.HostPipeline(eng))
# the pieces of a submit, in Python
s, c, q, t = batches[0]
for label, fn in (("join x3", lambda: ("".join(s).encode("ascii"), "".join(q).encode("ascii"), "".join(t).encode("ascii"))),
                  ("lengths x4", lambda: [np.fromiter(map(len, x), dtype=np.int32, count=len(x)) for x in (s, q, t)]),
                  ("ascontiguous", lambda: [np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 3) for x in c]),
                  ("concatenate", lambda: np.concatenate(c, axis=0)),
                  ("np.empty+touch", lambda: [np.empty((n, p.n_terms), dtype=np.float32).fill(0) for p in eng.predictors.values()])):
    t0 = time.perf_counter()
    fn()
    print(f"  {label:14s} {1e3 * (time.perf_counter() - t0):7.1f} ms")
