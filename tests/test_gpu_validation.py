"""The validation kit on the GPU box: the same self-tests as on CPU, now with the HIP path as one of the compared ways; and
the hook that pins the suite against a RELEASED model the day one is present (skipped until then)."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu
KIT = os.path.join(ROOT, "tests", "validation")


def test_validate_release_self_test_includes_the_hip_path(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(KIT, "validate_release.py"), "--self-test", "--golden-dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("[verdict] PASS") == 3
    assert r.stdout.count("[hip]   Predictor.forward_pass on cuda:0") == 3          # GCN, GCN + language model, CNN
    assert r.stdout.count(" graph vs hip") == 3 and r.stdout.count("oracle vs hip") == 3 and "EXCEEDS" not in r.stdout
    z = np.load(tmp_path / "release_bp.npz")
    assert np.max(np.abs(z["scores_hip"] - z["scores_graph"])) < 1e-4


def test_validate_opal_harness_on_the_gpu():
    r = subprocess.run([sys.executable, os.path.join(KIT, "validate_opal.py"), "--self-test", "--pairs", "60"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "oracle == HIP kernels" in r.stdout


def test_released_model_goldens_if_present():
    """tests/golden/release_<mode>.npz (written by validate_release.py next to a released .onnx, with onnxruntime's scores) +
    MDFRI_RELEASE_DIR pointing at the model files: the HIP path must reproduce onnxruntime's scores on the stored input within
    1e-4.  Nothing to check until such a fixture exists (no released file is available offline)."""
    fixtures = [f for f in glob.glob(os.path.join(GOLDEN, "release_*.npz")) if "scores_ort" in np.load(f).files]
    model_dir = os.environ.get("MDFRI_RELEASE_DIR")
    if not fixtures or not model_dir:
        pytest.skip("no released-model fixture with onnxruntime scores (tests/validation/validate_release.py MODEL.onnx --ort writes one)")
    from mDeepFRI.predict import Predictor
    import json
    for f in fixtures:
        z = np.load(f)
        rep = json.loads(bytes(z["report"]).decode())          # the fixture remembers which file it was made from and what kind it is
        name = os.path.basename(rep["file"])
        cand = glob.glob(os.path.join(model_dir, "**", name), recursive=True)
        assert cand, f"no {name} under {model_dir}"
        seq, cmap = bytes(z["seq"]).decode(), z["cmap"].astype(np.int32)
        is_cnn = (rep.get("mapped") or {}).get("kind") == "cnn"
        y = Predictor(cand[0]).forward_pass(seq) if is_cnn else Predictor(cand[0]).forward_pass(seq, cmap)
        assert np.max(np.abs(y - z["scores_ort"])) < 1e-4, name
