"""The validation kit (tests/validation/*.py) runs without a GPU: the file's own graph under ONNX semantics vs the oracle on
the mapped tensors; and the extra operators the NumPy ONNX runtime accepts for released (tf2onnx) graphs."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

KIT = os.path.join(ROOT, "tests", "validation")


def test_validate_release_self_test_runs_green_on_cpu(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(KIT, "validate_release.py"), "--self-test", "--golden-dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("[verdict] PASS") == 3 and "LSTM x2" in r.stdout and "Conv x4" in r.stdout
    z = np.load(tmp_path / "release_mf.npz")
    assert {"seq", "cmap", "scores_graph", "scores_oracle", "report"} <= set(z.files)


def test_validate_release_reports_an_unknown_graph_instead_of_guessing(tmp_path):
    """A graph that is not a DeepFRI model: the kit still executes it, says what it could not map, and does not call that a pass."""
    from mdfri_testkit import onnx_writer
    b = onnx_writer.GraphBuilder()
    x = b.node("MatMul", [b.input("seq"), b.const(np.ones((26, 4), np.float32))])
    b.output(b.node("Relu", [x]))
    p = tmp_path / "strange_mf.onnx"
    p.write_bytes(b.serialize())
    r = subprocess.run([sys.executable, os.path.join(KIT, "validate_release.py"), str(p), "--golden-dir", str(tmp_path)], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 2 and "[map]   FAILED" in r.stdout and "UNDECIDED" in r.stdout


def test_validate_opal_skips_cleanly_without_pyopal():
    r = subprocess.run([sys.executable, os.path.join(KIT, "validate_opal.py"), "--pairs", "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "[self-test]" in r.stdout or "[verdict] PASS" in r.stdout


def test_numpy_runtime_operators_of_converted_graphs():
    """tf2onnx leaves shape plumbing around Keras layers: Shape / Gather / Unsqueeze / Concat feeding Reshape, Cast, Slice, Where,
    Einsum, ReduceSum with the axes as attribute (opset < 13) or input, Squeeze without axes, LSTM with an initial state."""
    import onnx_numpy_runtime as rt
    from mDeepFRI import onnx_reader
    from mdfri_testkit import onnx_writer
    rng = np.random.default_rng(0)
    X = rng.standard_normal((1, 5, 6)).astype(np.float32)
    b = onnx_writer.GraphBuilder()
    x = b.input("x")
    shp = b.node("Shape", [x])
    n = b.node("Gather", [shp, b.const(np.array(1, np.int64))], axis=0)                      # L
    n1 = b.node("Unsqueeze", [n, b.const(np.array([0], np.int64))])
    target = b.node("Concat", [n1, b.const(np.array([-1], np.int64))], axis=0)              # (L, -1)
    flat = b.node("Reshape", [x, target])                                                   # (5, 6)
    sl = b.node("Slice", [flat, b.const(np.array([1], np.int64)), b.const(np.array([4], np.int64)), b.const(np.array([0], np.int64))])   # rows 1..3
    ein = b.node("Einsum", [sl, b.const(np.eye(6, dtype=np.float32) * 2)], equation=b"ij,jk->ik")
    mask = b.node("Greater", [ein, b.const(np.array(0, np.float32))])
    w = b.node("Where", [mask, ein, b.node("Neg", [ein])])                                   # |2x|
    rs = b.node("ReduceSum", [w], axes=[1], keepdims=0)                                      # attribute form
    rs2 = b.node("ReduceSum", [w, b.const(np.array([1], np.int64))], keepdims=0)             # input form
    b.output(b.node("Add", [rs, b.node("Cast", [rs2], to=1)]))
    g = onnx_reader.parse_model(b.serialize())
    got = rt.run(g, {"x": X})[0]
    np.testing.assert_allclose(got, 2 * np.abs(2 * X[0, 1:4].astype(np.float64)).sum(axis=1), rtol=1e-12)
    # LSTM with initial state == running the sequence in two halves
    H, T = 4, 6
    W, R, B = rng.standard_normal((1, 4 * H, 3)), rng.standard_normal((1, 4 * H, H)), rng.standard_normal((1, 8 * H))
    xs = rng.standard_normal((T, 1, 3))
    Y, Yh, Yc = rt._lstm(xs, W, R, B, H)
    Y1, h1, c1 = rt._lstm(xs[:3], W, R, B, H)
    Y2, h2, c2 = rt._lstm(xs[3:], W, R, B, H, h1, c1)
    np.testing.assert_allclose(np.concatenate([Y1, Y2]), Y, rtol=1e-12)
    np.testing.assert_allclose(h2, Yh, rtol=1e-12)
    with pytest.raises(NotImplementedError, match="Loop"):
        from mDeepFRI.onnx_reader import Graph, Node
        rt.run(Graph(nodes=[Node(op_type="Loop", name="l", inputs=[], outputs=["y"])], outputs=["y"]), {})


def test_validate_all_one_command_self_test(tmp_path):
    """tests/validation/validate_all.sh --self-test: the pin-day kit as one command, on this build's own exported graphs -- a GraphConv
    and a CNN file per {mf, bp, cc, ec} (cc and ec heads at their released sizes, 320 and 538 terms), then the aligner harness
    (PyOpal absent: skipped cleanly).  Every file PASSes, the table names the embedding variant, goldens are written per (kind, mode)."""
    r = subprocess.run(["bash", os.path.join(KIT, "validate_all.sh"), "--self-test", "--golden-dir", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    main, _, f16x3 = r.stdout.partition("--- the same files under MDFRI_HW_PIPE=f16x3")
    table = [ln for ln in main.splitlines() if ln.startswith("selftest-")]
    assert len(table) == 8 and all(" PASS " in ln for ln in table), table
    again = [ln for ln in f16x3.splitlines() if ln.startswith("selftest-")]      # the same files once more in a child under the opt-in pipe (informational)
    assert len(again) == 8 and all(" PASS " in ln for ln in again), again
    assert sum("embed_relu" in ln for ln in table) == 4 and sum("cnn 4 branches" in ln for ln in table) == 4
    assert "aligner (PyOpal / VTML80): rc=0" in r.stdout
    names = sorted(os.listdir(tmp_path))
    assert names == sorted(f"release_{k}_{m}.npz" for k in ("gcn", "cnn") for m in ("mf", "bp", "cc", "ec")), names
    z = np.load(tmp_path / "release_gcn_ec.npz")
    assert z["scores_graph"].shape == (538,) and np.max(np.abs(z["scores_graph"] - z["scores_oracle"])) < 1e-9
