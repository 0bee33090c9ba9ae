#!/usr/bin/env python3
"""Validate the MI355X path against a RELEASED model file -- the first thing to run on a box that has one.

    python tests/validation/validate_release.py DeepFRI-MERGED_GraphConv_gcd_512-512-512_fcd_1024_ca_10.0_mf.onnx [--ort]
    python tests/validation/validate_release.py --self-test          # on this build's own exported graphs (what CI runs)

Parity of the GCN / LSTM / CNN stages is "unpinned" offline: the arithmetic lives in the `.onnx` files of reference
mDeepFRI/__init__.py:47-80, executed by onnxruntime at reference mDeepFRI/predict.pyx:63-73,98 -- neither is in this image.
This script is the bridge for the day a file is present.  For one model file it

  1. parses it with mDeepFRI.onnx_reader (no `onnx` package needed) and prints the graph summary: inputs, outputs, operator
     histogram;
  2. maps its tensors onto the topology the HIP kernels implement (extract_weights) and prints what it mapped -- or the
     OnnxFormatError that says what it did not recognise;
  3. builds the reference's own test input -- a random 20-letter sequence and a dense 0/1 matrix
     `np.random.randint(0, 2, (L, L))` (reference weight_convert notebook, cell 1; fed as predict.pyx:82-90 feeds it);
  4. computes the scores four ways, each optional and reported separately:
       graph   the FILE'S OWN graph, executed operator by operator under ONNX semantics (tests/onnx_numpy_runtime.py, float64)
       ort     onnxruntime CPU on the file (`--ort`, or automatically when it imports): what the reference returns
       oracle  oracle/{gcn,lm,cnn}_oracle.py on the mapped tensors: is the restated topology the file's topology?
       hip     Predictor(weights=mapped).forward_pass on cuda:0: the product
  5. prints max |delta| for every pair against the north-star tolerance 1e-4, PASS / FAIL, and writes the input + all outputs
     to tests/golden/release_<mode>.npz (a fixture: data only) so that the GPU suite can pin against the released file from
     then on.

Exit code 0 = every available comparison within tolerance (and at least one referee of the file itself ran), 1 = a mismatch,
2 = the file could not be parsed / mapped / executed by any referee.  Test infrastructure: it imports oracle/ (the checker)."""
import argparse
import collections
import json
import os
import re
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "metagenomic-deepfri_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

TOL = 1e-4
AA20 = "ACDEFGHIKLMNPQRSTVWY"
ALPHABET = "-DGULNTKHYWCPVSOIEFXQABZRM"      # reference mDeepFRI/predict.pyx:26


def onehot(seq):
    S = np.zeros((len(seq), 26), dtype=np.float32)
    S[np.arange(len(seq)), [ALPHABET.index(c) for c in seq]] = 1
    return S


def mode_of(path):
    m = re.search(r"_(mf|bp|cc|ec)\.(onnx|mdfw|npz)$", os.path.basename(path))
    return m.group(1) if m else os.path.splitext(os.path.basename(path))[0]


def validate(path, length=300, seed=0, use_ort=None, device=0, out_dir=None, quiet=False, golden_name=None):
    from mDeepFRI import onnx_reader, weights as W
    import onnx_numpy_runtime
    say = (lambda *a: None) if quiet else print
    report = {"file": path, "mode": mode_of(path), "tolerance": TOL, "length": length, "seed": seed}
    # 1. parse
    try:
        graph = onnx_reader.parse_model(path)
    except Exception as e:
        say(f"[parse] FAILED: {type(e).__name__}: {e}")
        report["parse_error"] = f"{type(e).__name__}: {e}"
        return 2, report
    hist = collections.Counter(nd.op_type for nd in graph.nodes)
    report["graph"] = {"inputs": graph.inputs, "outputs": graph.outputs, "nodes": len(graph.nodes), "ops": dict(sorted(hist.items())),
                       "initializers": len(graph.initializers)}
    say(f"[parse] {os.path.basename(path)}: {len(graph.nodes)} nodes, {len(graph.initializers)} initializers; inputs {graph.inputs} -> outputs {graph.outputs}")
    say("[parse] operators: " + ", ".join(f"{k} x{v}" for k, v in sorted(hist.items())))
    # 2. map
    mapped, kind = None, None
    try:
        mapped = onnx_reader.extract_weights(graph)
        kind = W.model_kind(mapped)
        topo = W.validate_cnn(mapped) if kind == "cnn" else W.validate(mapped)
        report["mapped"] = {"kind": kind, "topology": {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in topo.items()},
                            "tensors": {k: list(np.asarray(v).shape) for k, v in mapped.items()}}
        say(f"[map]   kind={kind} topology={report['mapped']['topology']}")
        say("[map]   " + ", ".join(f"{k}{tuple(np.asarray(v).shape)}" for k, v in mapped.items()))
    except Exception as e:
        report["map_error"] = f"{type(e).__name__}: {e}"
        say(f"[map]   FAILED: {type(e).__name__}: {e}")
        say("[map]   the HIP path cannot load this file as is; the graph referee below still shows what the file computes")
    # 3. the reference's test input
    rng = np.random.RandomState(seed)                       # the notebook uses the legacy global generator
    seq = "".join(AA20[i] for i in rng.randint(0, 20, size=length))
    cmap = rng.randint(0, 2, size=(length, length)).astype(np.int32)
    S = onehot(seq)
    takes_cmap = len(graph.inputs) >= 2
    feeds = {graph.inputs[0]: cmap.reshape(1, length, length).astype(np.float32), graph.inputs[1]: S.reshape(1, length, 26)} if takes_cmap \
        else {graph.inputs[0]: S.reshape(1, length, 26)}
    results = {}
    # 4a. the file's own graph under ONNX semantics
    try:
        y = onnx_numpy_runtime.run(graph, feeds)[0]
        results["graph"] = np.asarray(y)[:, :, 0].reshape(-1)             # predict.pyx:100
        say(f"[graph] executed the file's graph in NumPy (float64): {results['graph'].shape[0]} terms")
    except Exception as e:
        report["graph_error"] = f"{type(e).__name__}: {e}"
        say(f"[graph] FAILED: {type(e).__name__}: {e}")
    # 4b. onnxruntime
    if use_ort is not False:
        try:
            import onnxruntime as rt
            so = rt.SessionOptions()
            so.intra_op_num_threads = so.inter_op_num_threads = 1
            sess = rt.InferenceSession(path, so, providers=["CPUExecutionProvider"])
            names = [i.name for i in sess.get_inputs()]
            results["ort"] = sess.run(None, {n: feeds[graph.inputs[k]] for k, n in enumerate(names)})[0][:, :, 0].reshape(-1)
            report["onnxruntime"] = rt.__version__
            say(f"[ort]   onnxruntime {rt.__version__} CPU: {results['ort'].shape[0]} terms")
        except ImportError as e:
            report["ort_error"] = f"ImportError: {e}"
            say("[ort]   onnxruntime is not installed here" + (" (--ort was requested!)" if use_ort else "; skipped"))
            if use_ort:
                return 2, report
        except Exception as e:
            report["ort_error"] = f"{type(e).__name__}: {e}"
            say(f"[ort]   FAILED: {type(e).__name__}: {e}")
    # 4c. the oracle on the mapped tensors
    if mapped is not None:
        try:
            if kind == "cnn":
                import cnn_oracle
                results["oracle"] = cnn_oracle.cnn_forward(mapped, seq, dtype=np.float64)
            elif "lm_U1" in mapped:
                import lm_oracle
                results["oracle"] = lm_oracle.gcn_lm_forward(mapped, seq, cmap, dtype=np.float64)
            else:
                import gcn_oracle
                results["oracle"] = gcn_oracle.gcn_forward(mapped, seq, cmap, dtype=np.float64)
            say(f"[oracle] CPU restatement on the mapped tensors (float64): {results['oracle'].shape[0]} terms")
        except Exception as e:
            report["oracle_error"] = f"{type(e).__name__}: {e}"
            say(f"[oracle] FAILED: {type(e).__name__}: {e}")
    # 4d. the HIP path
    if mapped is not None:
        try:
            from mDeepFRI import _hip
            if _hip.device_count() <= 0:
                raise RuntimeError("no HIP device visible")
            from mDeepFRI.predict import Predictor
            pred = Predictor(path, weights=mapped, device=device)
            results["hip"] = pred.forward_pass(seq, cmap) if takes_cmap else pred.forward_pass(seq)
            say(f"[hip]   Predictor.forward_pass on cuda:{device}: {results['hip'].shape[0]} terms")
        except Exception as e:
            report["hip_error"] = f"{type(e).__name__}: {e}"
            say(f"[hip]   skipped: {type(e).__name__}: {e}")
    # 5. verdict
    names = [k for k in ("ort", "graph", "oracle", "hip") if k in results]
    deltas, worst = {}, 0.0
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            if results[a].shape != results[b].shape:
                deltas[f"{a}-{b}"] = None
                worst = float("inf")
                say(f"[delta] {a:>6} vs {b:<6}: SHAPES DIFFER {results[a].shape} vs {results[b].shape}")
                continue
            d = float(np.max(np.abs(np.asarray(results[a], dtype=np.float64) - np.asarray(results[b], dtype=np.float64))))
            deltas[f"{a}-{b}"] = d
            worst = max(worst, d)
            say(f"[delta] {a:>6} vs {b:<6}: max |delta| = {d:.3e}   {'ok' if d < TOL else 'EXCEEDS 1e-4'}")
    report["max_abs_delta"] = deltas
    file_referee = "ort" in results or "graph" in results
    if not file_referee:
        say("[verdict] UNDECIDED: neither onnxruntime nor the NumPy runtime could execute the file's graph")
        return 2, report
    if len(names) < 2:
        say("[verdict] UNDECIDED: only one way of computing the scores was available")
        return 2, report
    ok = worst < TOL
    report["verdict"] = "PASS" if ok else "FAIL"
    say(f"[verdict] {'PASS' if ok else 'FAIL'}: {len(deltas)} comparison(s) among {names}, worst {worst:.3e} against {TOL:g}")
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        out = os.path.join(out_dir, f"release_{golden_name or report['mode']}.npz")
        np.savez_compressed(out, seq=np.frombuffer(seq.encode(), dtype=np.uint8), cmap=cmap.astype(np.uint8),
                            report=np.frombuffer(json.dumps(report).encode(), dtype=np.uint8), **{f"scores_{k}": np.asarray(v) for k, v in results.items()})
        say(f"[golden] wrote {out} (sequence, map, scores of {names})")
    return (0 if ok else 1), report


def self_test(out_dir=None, quiet=False):
    """The kit on this build's own exported files (mdfri_testkit.onnx_writer): GCN, GCN + language model, sequence-only CNN.  What
    this proves: the kit runs end to end; the exported graphs, executed under ONNX semantics, agree with the oracles and -- on a
    GPU box -- with the HIP path.  What it cannot prove: anything about a released file."""
    from mdfri_testkit import onnx_writer, synthetic
    tmp = tempfile.mkdtemp(prefix="mdfri_validate_")
    w_gcn = synthetic.glorot_gcn_weights(seed=3, n_terms=37, embed=256, gc_dims=(256, 256, 256), fc_dim=256)
    w_lm = dict(synthetic.glorot_gcn_weights(seed=4, n_terms=21, embed=256, gc_dims=(256, 256), fc_dim=256))
    w_lm.update(synthetic.glorot_lm_weights(seed=5, hidden=64, embed=256))
    w_cnn = synthetic.glorot_cnn_weights(seed=6, n_terms=19)
    files = {"selftest-gcn_mf.onnx": onnx_writer.deepfri_gcn_model(w_gcn, raw=False, use_gemm_head=True),
             "selftest-gcnlm_bp.onnx": onnx_writer.deepfri_gcn_model(w_lm),
             "selftest-cnn_cc.onnx": onnx_writer.deepcnn_model(w_cnn)}
    worst = 0
    for name, blob in files.items():
        path = os.path.join(tmp, name)
        open(path, "wb").write(blob)
        if not quiet:
            print(f"=== {name} ===")
        rc, rep = validate(path, length=120, seed=1, use_ort=None, out_dir=out_dir, quiet=quiet)
        worst = max(worst, rc)
    return worst


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("model", nargs="?", help="path of a .onnx model file")
    ap.add_argument("--self-test", action="store_true", help="run on this build's own exported graphs")
    ap.add_argument("--ort", action="store_true", help="require onnxruntime (default: use it when it imports)")
    ap.add_argument("--no-ort", action="store_true", help="do not try onnxruntime")
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--golden-dir", default=None, help="where to write release_<mode>.npz (default for a real file: tests/golden/)")
    ap.add_argument("--json", action="store_true", help="print the report as one JSON line at the end")
    args = ap.parse_args(argv)
    if args.self_test:
        return self_test(out_dir=args.golden_dir)
    if not args.model:
        ap.error("a model file (or --self-test) is required")
    rc, rep = validate(args.model, args.length, args.seed, use_ort=(True if args.ort else False if args.no_ort else None), device=args.device,
                       out_dir=args.golden_dir or os.path.join(ROOT, "tests", "golden"))
    if args.json:
        print(json.dumps(rep))
    return rc


if __name__ == "__main__":
    sys.exit(main())
