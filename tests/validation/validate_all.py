#!/usr/bin/env python3
"""Driver behind validate_all.sh: every released model file of a directory through validate_release.validate(), then the aligner
through validate_opal, one summary table.  See validate_all.sh for the contract.  Test infrastructure."""
import argparse
import glob
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)

import validate_release as vr  # noqa: E402  (sets up sys.path for the package, the oracle and tests/)


def kind_of(path):
    b = os.path.basename(path).lower()
    return "cnn" if "cnn" in b else "gcn"


def variant_of(report):
    """Which of the topology variants that only a released file can settle the graph turned out to be."""
    m = report.get("mapped")
    if not m:
        return "unmapped: " + report.get("map_error", report.get("parse_error", "?"))[:60]
    if m["kind"] == "cnn":
        t = m["topology"]
        return f"cnn {len(t.get('kernel_lens', []))} branches"
    t, names = m["topology"], set(m["tensors"])
    bits = ["lm" if t.get("lm_dim") else "no-lm", "embed_linear" if "embed_linear" in names else "embed_relu"]
    if "b_aa" in names:
        bits.append("b_aa")
    if t.get("embed") == 26:
        bits.append("identity-embedding")
    return "gcn " + "+".join(bits)


def self_test_files(tmp):
    """This build's own exported graphs, one per (kind, mode) incl. an EC-sized head (T = 538): what the kit runs on today."""
    from mdfri_testkit import onnx_writer, synthetic
    small = dict(embed=256, gc_dims=(256, 256, 256), fc_dim=256)
    out = []
    for i, mode in enumerate(("mf", "bp", "cc", "ec")):
        T = synthetic.GO_TERMS[mode] if mode in ("cc", "ec") else 40 + i    # two heads at their released size (320, 538), two small (time)
        w = synthetic.glorot_gcn_weights(seed=20 + i, n_terms=T, **small)
        p = os.path.join(tmp, f"selftest-GraphConv_{mode}.onnx")
        open(p, "wb").write(onnx_writer.deepfri_gcn_model(w, raw=(i % 2 == 0), use_gemm_head=(i % 2 == 1)))
        out.append(p)
        wc = synthetic.glorot_cnn_weights(seed=30 + i, n_terms=T if mode == "ec" else 17 + i)
        p = os.path.join(tmp, f"selftest-CNN_{mode}.onnx")
        open(p, "wb").write(onnx_writer.deepcnn_model(wc))
        out.append(p)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model_dir", nargs="?")
    ap.add_argument("--self-test", action="store_true")
    ap.add_argument("--length", type=int, default=None, help="test sequence length (default 300; 120 for --self-test)")
    ap.add_argument("--golden-dir", default=None, help="where release_<kind>_<mode>.npz go (default tests/golden/; a temp dir for --self-test)")
    ap.add_argument("--no-opal", action="store_true")
    args = ap.parse_args()
    have_ort = importlib.util.find_spec("onnxruntime") is not None
    if args.self_test:
        tmp = tempfile.mkdtemp(prefix="mdfri_validate_all_")
        files, golden, length = self_test_files(tmp), args.golden_dir or os.path.join(tmp, "golden"), args.length or 120
    else:
        if not args.model_dir or not os.path.isdir(args.model_dir):
            ap.error("MODEL_DIR (a directory with the released .onnx files) or --self-test is required")
        files = sorted(f for f in glob.glob(os.path.join(args.model_dir, "**", "*.onnx"), recursive=True) if re.search(r"_(mf|bp|cc|ec)\.onnx$", f))
        golden, length = args.golden_dir or os.path.join(ROOT, "tests", "golden"), args.length or 300
        if not files:
            print(f"no *_{{mf,bp,cc,ec}}.onnx file under {args.model_dir}")
            return 2
    rows, worst = [], 0
    for f in files:
        print(f"=== {os.path.basename(f)} ===", flush=True)
        # goldens are named by kind AND mode: a model directory holds a GraphConv and a CNN file per mode
        rc, rep = vr.validate(f, length=length, seed=0, use_ort=(True if have_ort else None), out_dir=golden, golden_name=f"{kind_of(f)}_{vr.mode_of(f)}")
        worst = max(worst, rc)
        deltas = [d for d in (rep.get("max_abs_delta") or {}).values() if d is not None]
        rows.append((os.path.basename(f), kind_of(f), rep.get("mode"), variant_of(rep), rep.get("verdict", "UNDECIDED" if rc == 2 else "FAIL"),
                     f"{max(deltas):.2e}" if deltas else "-", ",".join(k.split("-")[0] for k in (rep.get("max_abs_delta") or {})) or "-"))
    opal = "skipped (--no-opal)"
    opal_rc = 0
    if not args.no_opal:
        cmd = [sys.executable, os.path.join(HERE, "validate_opal.py")] + (["--self-test"] if args.self_test else [])
        r = subprocess.run(cmd, capture_output=True, text=True)
        opal_rc = r.returncode
        tail = [ln for ln in r.stdout.splitlines() if ln.strip()]
        hit = [ln for ln in tail if "TIE_RULE" in ln or "not installed" in ln.lower() or "absent" in ln.lower() or "self-test" in ln.lower()]
        opal = (hit[-1] if hit else (tail[-1] if tail else r.stderr.strip()[-200:])).strip()
    print("\n" + "=" * 118)
    print(f"{'file':44s} {'kind':4s} {'mode':4s} {'variant':34s} {'verdict':9s} {'worst':>9s}")
    for r in rows:
        print(f"{r[0][:44]:44s} {r[1]:4s} {str(r[2]):4s} {r[3][:34]:34s} {r[4]:9s} {r[5]:>9s}")
    print(f"onnxruntime: {'used' if have_ort else 'not importable here (graph referee = tests/onnx_numpy_runtime.py)'}")
    print(f"aligner (PyOpal / VTML80): rc={opal_rc}: {opal}")
    print(f"goldens: {golden}")
    # The same files once more under the opt-in pipe (include/mdfri.h: MDFRI_HW_PIPE=f16x3 -- its activation scale is a constant, DESIGN.md section 4:
    # this run tells whether the released weights stay inside its range).  A child process (the pipe is read once per process), goldens into a
    # scratch directory, no aligner check; informational: its verdicts are printed, the exit code is the default pipe's.
    if os.environ.get("MDFRI_HW_PIPE", "") == "" and os.environ.get("MDFRI_VALIDATE_CHILD") is None:
        env = dict(os.environ, MDFRI_HW_PIPE="f16x3", MDFRI_VALIDATE_CHILD="1")
        cmd = [sys.executable, os.path.abspath(__file__)] + (["--self-test"] if args.self_test else [args.model_dir]) + \
              ["--no-opal", "--golden-dir", tempfile.mkdtemp(prefix="mdfri_validate_f16x3_"), "--length", str(length)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        table = r.stdout.split("=" * 118)[-1].strip() if "=" * 118 in r.stdout else (r.stdout.strip()[-600:] or r.stderr.strip()[-600:])
        print("\n--- the same files under MDFRI_HW_PIPE=f16x3 (opt-in pipe; informational, rc=%d) ---" % r.returncode)
        print(table)
    return max(worst, 1 if opal_rc == 1 else 0)


if __name__ == "__main__":
    sys.exit(main())
