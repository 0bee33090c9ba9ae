"""EXPERIMENT driver for csrc/tools/l1ax_probe.hip: layer 1 + layer-2 aggregation as one matrix-pipe kernel on the REAL adjacency of a
configs[2] chunk (128 synthetic L=512 proteins = 65 536 rows).  Builds the per-group records on the host from the CSR the library
produced, runs the kernel, checks it against float64 (Z2 = Ahat . elu(S . T1) and the per-group pool partials), prints microseconds."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
import torch  # noqa: E402
from mDeepFRI import synthetic  # noqa: E402
from mDeepFRI.batch import HotPathEngine, PackedProteins  # noqa: E402
from mDeepFRI.predict import Predictor  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "metagenomic-deepfri_amd", "lib", os.environ.get("LXP_LIB", "libl1ax_probe.so")))
L = int(os.environ.get("AXL", 512))
prots = synthetic.synthetic_proteins(3, 65536 // L, L, coords=os.environ.get("LXP_COORDS", "walk"))
pred = Predictor("probe", weights=synthetic.glorot_gcn_weights(seed=0, n_terms=16))
eng = HotPathEngine({"mf": pred}, max_rows=65536)
pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots], max_rows=65536)
db = eng.upload(pk)
eng.forward_alignments(db)
eng.check(db)
R = pk.chunks[0].rows
rowptr = eng._bufs["rowptr"][:R + 1].cpu().numpy().astype(np.int64)
nnz = int(rowptr[-1])
colidx = eng._bufs["colidx"][:nnz].cpu().numpy()
val = eng._bufs["val"][:nnz].cpu().numpy()
S = eng._bufs["lsum"][:R * 32].cpu().numpy().reshape(R, 32).copy()      # the chunk's real letter sums
rng = np.random.default_rng(0)
T1 = (rng.standard_normal((32, 512)) * 0.3).astype(np.float32)
T1[26:] = 0
UMAX, RF = lib.l1ax_umax(), lib.l1ax_rec_floats()
G = R // 32
rec = np.zeros((G, RF), np.float32)
hdr = rec[:, :16].view(np.int32)
own = rec[:, 16:16 + UMAX]
what = rec[:, 16 + UMAX:16 + UMAX + 32 * UMAX].reshape(G, UMAX, 32)
su = rec[:, 16 + UMAX + 32 * UMAX:].reshape(G, UMAX, 32)
rows_of = np.repeat(np.arange(R), np.diff(rowptr))
skipped, Us = [], []
for g in range(G):
    e0, e1 = rowptr[g * 32], rowptr[g * 32 + 32]
    u, inv = np.unique(colidx[e0:e1], return_inverse=True)
    if len(u) > UMAX:
        skipped.append(g)
        continue
    hdr[g, 0] = len(u)
    own[g, :len(u)] = ((u >= g * 32) & (u < g * 32 + 32)).astype(np.float32)
    what[g, inv, rows_of[e0:e1] - g * 32] = val[e0:e1]
    su[g, :len(u)] = S[u]
    Us.append(len(u))
print(f"{G} groups, {nnz / R:.2f} entries/row, union per group: mean {np.mean(Us):.1f} max {max(Us)}; UMAX {UMAX}: {len(skipped)} outlier groups skipped")
dev = torch.device("cuda:0")
recs, t1d = torch.from_numpy(rec).to(dev), torch.from_numpy(T1).to(dev)
out = torch.zeros(R, 512, device=dev)
pool = torch.zeros(G, 512, device=dev)
lib.l1ax_run.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_void_p]
st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
run = lambda: lib.l1ax_run(recs.data_ptr(), t1d.data_ptr(), out.data_ptr(), pool.data_ptr(), G, st)  # noqa: E731
assert run() == 0
torch.cuda.synchronize()
x = S.astype(np.float64) @ T1.astype(np.float64)
H1 = np.where(x > 0, x, np.exp(x) - 1)
oh, ph = out.cpu().numpy(), pool.cpu().numpy()
worst = worst_p = 0.0
for row in list(range(64)) + list(rng.integers(0, R, size=300)):
    if row // 32 in skipped:
        continue
    e0, e1 = rowptr[row], rowptr[row + 1]
    worst = max(worst, float(np.abs(oh[row] - (val[e0:e1, None].astype(np.float64) * H1[colidx[e0:e1]]).sum(0)).max()))
for g in list(range(4)) + list(rng.integers(0, G, size=60)):
    if g in skipped:
        continue
    worst_p = max(worst_p, float(np.abs(ph[g] - H1[g * 32:g * 32 + 32].sum(0)).max()))
print(f"max |Z2 - float64| over 364 rows: {worst:.3e}; max |pool partial - float64| over 64 groups: {worst_p:.3e}")
assert (worst < 1e-4 and worst_p < 1e-4) or "abl" in os.environ.get("LXP_LIB", "")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    run()
a.record()
for _ in range(20):
    run()
b.record()
torch.cuda.synchronize()
print(f"fused layer 1 + layer-2 aggregation: {a.elapsed_time(b) * 1e3 / 20:.2f} us/launch  (today: k_gemm_f32<L1> ~41 us + k_aggregate ~68 us)")
