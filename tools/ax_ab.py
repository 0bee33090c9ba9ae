"""Round 6: A/B of two (or more) builds of the library on the GraphConv launches.  Each library runs in a process of its own
(`MDFRI_HIP_LIB`), on the same synthetic workloads at the default chunk; prints the mean HIP-event time per launch of the two
aggregation classes (ax2 = the layer-2 launch that makes layer 1, ax3 = layer 3) and of the two products, the step time, and a sha256 of
all scores -- equal digests = bit-identical builds.
    python tools/ax_ab.py experiments/_r06/base/libmdfri_hip.so metagenomic-deepfri_amd/lib/libmdfri_hip.so
"""
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def worker():
    sys.path.insert(0, os.path.join(ROOT, "metagenomic-deepfri_amd"))
    sys.path.insert(0, ROOT)
    import numpy as np  # noqa: F401
    import torch
    from mDeepFRI import _hip
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    from mdfri_testkit import synthetic

    import ctypes
    probe = ctypes.CDLL(_hip.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name in [n for n in _hip.SIGNATURES if not hasattr(probe, n)]:   # an older build of the library: entry points it does not have yet are not used here
        del _hip.SIGNATURES[name]
    lib = _hip.lib()
    heads = {k: Predictor("ab-" + k, weights=synthetic.glorot_gcn_weights(seed=i, n_terms=synthetic.GO_TERMS[k])) for i, k in enumerate(("mf", "bp", "cc"))}
    eng = HotPathEngine(heads)
    h = hashlib.sha256()
    res = {}
    shapes = (("L512", 3, 1024, 512, 0.0), ("L256", 4, 2048, 256, 0.0), ("L1024", 5, 512, 1024, 0.0), ("mixed", 6, 1200, (128, 1024), 0.05),
              ("L448", 7, 1024, 448, 0.03), ("L200", 8, 2048, 200, 0.0))
    only = os.environ.get("AX_AB_SHAPES")
    if only:      # any fixed length: "L384" = one chunk's worth of 384-residue proteins (x 2)
        known = {sh[0]: sh for sh in shapes}
        shapes = [known[nm] if nm in known else (nm, 100 + int(nm[1:]), max(2, 2 * 262144 // int(nm[1:])), int(nm[1:]), 0.0) for nm in only.split(",")]
    for name, seed, n, length, indel in shapes:
        prots = synthetic.synthetic_proteins(seed, n, length, indel)
        pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots])
        db = eng.upload(pk)
        for _ in range(2):
            out = eng.forward_alignments(db)
        eng.check(db)
        torch.cuda.synchronize()
        lib.mdf_timing_reset()
        lib.mdf_timing_enable(1)
        reps = 6
        t0 = time.perf_counter()
        for _ in range(reps):
            out = eng.forward_alignments(db)
        eng.check(db)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        lib.mdf_timing_enable(0)
        row = {"ms_per_pass": round(1e3 * dt, 3), "proteins_per_s": round(n / dt, 1)}
        for cls in ("ax2", "ax3", "gemm2", "gemm3", "cmap", "gemm1"):
            cnt, ms = _hip.c_int64(0), _hip.ctypes.c_double(0.0)
            lib.mdf_timing_read(cls.encode(), cnt, ms)
            row[cls] = round(1e3 * ms.value / max(cnt.value, 1), 2)
        for k in ("mf", "bp", "cc"):
            h.update(out[k].cpu().numpy().tobytes())
        res[name] = row
    res["sha256"] = h.hexdigest()[:16]
    res["version"] = lib.mdf_version().decode()
    print("AXAB " + json.dumps(res), flush=True)


if __name__ == "__main__":
    if os.environ.get("AX_AB_WORKER"):
        worker()
        sys.exit(0)
    libs = sys.argv[1:]
    rounds = int(os.environ.get("AX_AB_ROUNDS", "2"))
    rows = {}
    for r in range(rounds):      # interleaved: the box's clock drifts with temperature and power
        for lb in libs:
            env = dict(os.environ, AX_AB_WORKER="1", MDFRI_HIP_LIB=os.path.abspath(lb))
            out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith("AXAB ")]
            if not line:
                print(lb, "FAILED", out.stdout[-2000:], out.stderr[-4000:])
                continue
            rows.setdefault(lb, []).append(json.loads(line[0][5:]))
    for lb, rs in rows.items():
        print("==", lb, rs[0]["version"], "sha256", [x["sha256"] for x in rs])
        for name in rs[0]:
            if name in ("sha256", "version"):
                continue
            print(f"  {name:6s} " + "  ".join(f"{k} " + "/".join(f"{x[name][k]:.1f}" for x in rs) for k in rs[0][name]))
