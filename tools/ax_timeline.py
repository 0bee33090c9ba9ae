"""Round 6: where a wave of the matrix-pipe aggregation kernel spends its life.  Needs the probe build of the library
(experiments/r06_build_probe.sh -> experiments/_r06/probe/libmdfri_hip.so: every wave stamps the 100 MHz wall clock at the kernel's phase
boundaries); runs one forward pass of a single GO head over proteins of one length and prints, per launch kind (the stamps of the LAST launch
of each kind survive: layer 3 overwrites layer 2 -- so `--layer 2` stops after the layer-2 aggregation by asking for a model with two layers),
the mean time between consecutive stamps.
    MDFRI_HIP_LIB=experiments/_r06/probe/libmdfri_hip.so python tools/ax_timeline.py --length 512 --layers 3
"""
import argparse
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from mDeepFRI import _hip  # noqa: E402
from mDeepFRI.batch import HotPathEngine, PackedProteins  # noqa: E402
from mDeepFRI.predict import Predictor  # noqa: E402
from mdfri_testkit import synthetic  # noqa: E402

NAMES = {0: "start", 29: "before-last-barrier", 30: "after-last-barrier", 31: "stores-issued"}
for c in range(4):
    for k, n in enumerate(("loop-top", "tile/requests done", "after B1", "split done", "after B2", "matrix done")):
        NAMES[1 + 6 * c + k] = f"c{c} {n}"

ap = argparse.ArgumentParser()
ap.add_argument("--length", type=int, default=512)
ap.add_argument("--proteins", type=int, default=512)
ap.add_argument("--layers", type=int, default=3, help="2: the last aggregation launch is the layer-2 one (layer 1 inside); 3: layer 3 (plain form)")
a = ap.parse_args()
lib = _hip.lib()
lib.mdf_debug_ax_probe.argtypes = [ctypes.c_void_p]
lib.mdf_debug_ax_probe.restype = ctypes.c_int
gc = (512, 512, 512)[: a.layers]
w = synthetic.glorot_gcn_weights(seed=0, n_terms=489, gc_dims=gc)
eng = HotPathEngine({"mf": Predictor("tl", weights=w)})
prots = synthetic.synthetic_proteins(3, a.proteins, a.length)
pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots])
db = eng.upload(pk)
for _ in range(2):
    eng.forward_alignments(db)
torch.cuda.synchronize()
n_wg = (a.proteins + 7) // 8 * 8 * 16      # (the grid is padded to whole groups of eight proteins: XCD-aware order)
buf = torch.zeros((n_wg, 8, 32), dtype=torch.int64, device="cuda")
assert lib.mdf_debug_ax_probe(buf.data_ptr()) == 0
eng.forward_alignments(db)
torch.cuda.synchronize()
assert lib.mdf_debug_ax_probe(None) == 0
t = buf.cpu().numpy().astype(np.float64)
valid = t[:, :, 0] > 0
t0 = t[:, :, 0].copy()
print(f"length {a.length}, {a.proteins} proteins, {a.layers} layers; workgroups with stamps: {int(valid[:, 0].sum())} of {n_wg}")
start = t0[valid].min()
end = t[:, :, 31][valid].max()
print(f"launch span by the stamps: {(end - start) / 100:.1f} us; wave life mean {(t[:, :, 31][valid] - t0[valid]).mean() / 100:.2f} us")
prev = 0
for k in range(1, 32):
    col = t[:, :, k]
    ok = valid & (col > 0)
    if not ok.any():
        continue
    d = (col[ok] - t[:, :, prev][ok]) / 100.0
    print(f"  {NAMES[prev]:24s} -> {NAMES[k]:24s} mean {d.mean():6.2f} us   p10 {np.percentile(d, 10):6.2f}  p90 {np.percentile(d, 90):6.2f}   waves {int(ok.sum())}")
    prev = k
# how many workgroups were alive at once (by start/end of wave 0): concurrency check
ev = sorted([(x, 1) for x in t0[:, 0][valid[:, 0]]] + [(x, -1) for x in t[:, 0, 31][valid[:, 0]]])
cur = peak = 0
for _, s in ev:
    cur += s
    peak = max(peak, cur)
print("peak concurrent workgroups:", peak, "(256 CUs)")
