"""Rate of the two text formatters of the output stage (host code of the library; no GPU needed): the prediction matrix
(`mdf_matrix_format_host`: repr of every score, reference pipeline.py:318-319) by thread count, against Python's csv.writer, and the
results.tsv lines (`mdf_results_format_host`, pipeline.py:713-716) against per-line Python formatting."""
import csv
import io
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "metagenomic-deepfri_amd"))
from mDeepFRI.output import prediction_matrix_text, results_text, write_prediction_matrix  # noqa: E402

B, T = int(os.environ.get("FB", 4000)), 2752
rng = np.random.default_rng(0)
s = (rng.random((B, T)) ** 8).astype(np.float32)
ids = [f"protein_{i}" for i in range(B)]
for th in (1, 4, 16, 32, 0):
    best = 1e9
    for _ in range(3):
        with open(os.devnull, "wb") as fh:
            t0 = time.perf_counter()
            nbytes = write_prediction_matrix(fh, ids, s, "gcn", threads=th)
            best = min(best, time.perf_counter() - t0)
    print(f"prediction matrix, {B} x {T} scores, threads={th or 'auto'}: {best * 1e3:7.1f} ms = {B / best / 1e3:6.1f} k proteins/s, {s.size / best / 1e6:6.1f} M scores/s, {nbytes / 1e6:.0f} MB")
t0 = time.perf_counter()
buf = io.StringIO()
w = csv.writer(buf, delimiter="\t")
for q, row in zip(ids[:400], s[:400]):
    w.writerow([q, "gcn"] + row.tolist())
dt = (time.perf_counter() - t0) * B / min(400, B)
print(f"the reference's way (tolist + csv.writer, one core): {dt * 1e3:7.1f} ms = {B / dt / 1e3:6.1f} k proteins/s")
assert buf.getvalue() == prediction_matrix_text(ids[:400], s[:400], "gcn").decode()
off = np.zeros(B + 1, np.int32)
keep = [np.flatnonzero(r >= 0.1) for r in s]
off[1:] = np.cumsum([len(k) for k in keep])
ti = np.concatenate(keep).astype(np.int32)
sc = np.concatenate([r[k] for r, k in zip(s, keep)]).astype(np.float32)
terms = [f"GO:{k:07d}" for k in range(T)]
t0 = time.perf_counter()
text = results_text(ids, "gcn", "Molecular Function", terms, terms, off, ti, sc)
dt = time.perf_counter() - t0
t0 = time.perf_counter()
n = 0
for p in range(min(400, B)):
    for k in range(off[p], off[p + 1]):
        n += len(f"{ids[p]}\tgcn\tMolecular Function\t{terms[ti[k]]}\t{float(sc[k]):.4f}\t{terms[ti[k]]}\tnan\tnan\tnan\tnan\tnan\tnan\n")
dp = (time.perf_counter() - t0) * B / min(400, B)
print(f"results.tsv lines: {len(ti)} lines in {dt * 1e3:.1f} ms ({len(ti) / dt / 1e6:.1f} M lines/s); per-line Python formatting: {dp * 1e3:.1f} ms")
