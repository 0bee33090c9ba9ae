"""The compiled reference-side binding (tests/binding/*.pyx -> Cython -> g++, linked with libmdfri_hip.so) on the GPU: the
reference's own known-answer tests re-expressed (tests/test_contact_map_utils.py:15-25, tests/test_predict.py:9-33), the golden
align cases, and the reference's forked-Pool call pattern for build_align_contact_map (pipeline.py:476-481)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import cmap_oracle as orc
import gcn_oracle
from conftest import ROOT, gstr
from mDeepFRI import weights
from mdfri_testkit import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def compiled():
    import binding_loader
    return binding_loader.load()


def test_pairwise_sqeuclidean_reference_kat_and_goldens(compiled, cmap_golden):
    cmu, _ = compiled
    np.random.seed(42)                                  # reference tests/test_contact_map_utils.py:15-25
    result = cmu.pairwise_sqeuclidean(np.random.rand(3, 3).astype(np.float32))
    expected = np.array([[0, 1.01354558, 0.12442072], [1.01354558, 0, 0.99467713], [0.12442072, 0.99467713, 0]], dtype=np.float32)
    assert np.allclose(result, expected)
    rng = np.random.default_rng(5)
    for n in (1, 64, 300):
        X = synthetic.random_walk_coords(rng, n)
        D = cmu.pairwise_sqeuclidean(X, threads=1)
        assert D.dtype == np.float32 and np.array_equal(D.view(np.uint32), orc.pairwise_sqeuclidean(X).view(np.uint32))   # bit patterns


def test_align_contact_map_golden_cases(compiled, cmap_golden):
    cmu, _ = compiled
    names = [str(x) for x in cmap_golden["index/align"]]
    assert len(names) >= 20
    for n in names:
        out = cmu.align_contact_map(gstr(cmap_golden[n + "/q"]), gstr(cmap_golden[n + "/t"]), cmap_golden[n + "/pairs"],
                                    int(cmap_golden[n + "/gen"]))
        assert out.dtype == np.int32 and np.array_equal(out, cmap_golden[n + "/out"]), n
    assert np.array_equal(cmu.align_contact_map("AB", "AB", np.array([[0, 1]], dtype=np.int32)), [[1, 1], [0, 1]])   # the .pyx-actual output


def test_seq2onehot_reference_kats(compiled):
    _, pr = compiled                                    # reference tests/test_predict.py:9-33
    r = pr.seq2onehot("")
    assert r.shape == (0, 26) and r.dtype == np.float32
    exp = np.zeros((4, 26), np.float32)
    exp[0, 21] = exp[1, 11] = exp[2, 1] = exp[3, 17] = 1
    assert np.array_equal(pr.seq2onehot("ACDE"), exp)
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        pr.seq2onehot("AJD*Z")
    with pytest.raises(ValueError, match="Invalid character in sequence: a"):
        pr.seq2onehot("ACDa")
    assert np.array_equal(pr.seq2onehot("-DGULNTKHYWCPVSOIEFXQABZRM"), np.eye(26, dtype=np.float32))


def test_predictor_gcn_lm_and_cnn_through_the_compiled_class(compiled, tmp_path):
    _, pr = compiled
    from mDeepFRI.predict import Predictor as CtypesPredictor
    rng = np.random.default_rng(3)
    seq = synthetic.random_sequence(rng, 140)
    cm = orc.calculate_contact_map(synthetic.random_walk_coords(rng, 140), 6.0)
    w = synthetic.glorot_gcn_weights(seed=4, n_terms=33)
    path = tmp_path / "DeepFRI-SYNTH_GraphConv_gcd_512-512-512_fcd_1024_ca_10.0_mf.mdfw"
    weights.save_mdfw(str(path), w)
    p = pr.Predictor(str(path).replace(".mdfw", ".onnx"), threads=1)      # the pipeline passes the .onnx name (pipeline.py:584)
    assert p.model_path.endswith(".onnx") and p.threads == 1 and p.input_names == ["cmap", "seq"] and p.session is not None
    y = p.forward_pass(seq, cm)
    assert y.dtype == np.float32 and y.shape == (33,)
    assert np.max(np.abs(y - gcn_oracle.gcn_forward(w, seq, cm))) < 1e-4
    assert np.array_equal(y, CtypesPredictor("x", weights=w).forward_pass(seq, cm))      # both bindings, one library: identical
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        p.forward_pass("AJD*Z", np.eye(5, dtype=np.int32))
    # language-model branch and the sequence-only CNN (predict.pyx:91-95: cmap=None)
    import cnn_oracle
    import lm_oracle
    wl = synthetic.glorot_gcn_weights(seed=1, n_terms=20, embed=256, gc_dims=(256, 256), fc_dim=256)
    wl.update(synthetic.glorot_lm_weights(seed=2, hidden=64, embed=256))
    weights.save_mdfw(str(tmp_path / "lm.mdfw"), wl)
    yl = pr.Predictor(str(tmp_path / "lm.mdfw")).forward_pass(seq, cm)
    assert np.max(np.abs(yl - lm_oracle.gcn_lm_forward(wl, seq, cm))) < 1e-4
    wc = synthetic.glorot_cnn_weights(seed=3, n_terms=20)
    weights.save_mdfw(str(tmp_path / "DeepCNN-SYNTH_mf.mdfw"), wc)
    pc = pr.Predictor(str(tmp_path / "DeepCNN-SYNTH_mf.onnx"))
    assert pc.input_names == ["seq"]
    assert np.max(np.abs(pc.forward_pass(seq) - cnn_oracle.cnn_forward(wc, seq))) < 1e-4


def test_predict_batch_10k_l512_through_the_compiled_binding(compiled, tmp_path):
    """BASELINE configs[2]'s batch -- 10 000 synthetic L=512 proteins, MF + CC heads -- through the COMPILED binding alone:
    `predict_batch` is one cdef-extern call into mdf_engine_run_alignments_host (no torch, no ctypes, no Python loop over chunks).
    Bitwise equal to the ctypes/torch engine on the same inputs; a sample is checked against the per-call compiled
    `Predictor.forward_pass` (bitwise) and the oracle (1e-4)."""
    _, pr = compiled
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor as CtypesPredictor
    seqs, coords, q_alns, t_alns = bench.make_fixed_length(42 + 2, 10_000, 512)
    ws = {"mf": synthetic.glorot_gcn_weights(seed=0, n_terms=synthetic.GO_TERMS["mf"]), "cc": synthetic.glorot_gcn_weights(seed=2, n_terms=synthetic.GO_TERMS["cc"])}
    preds = []
    for m, w in ws.items():
        weights.save_mdfw(str(tmp_path / f"{m}.mdfw"), w)
        preds.append(pr.Predictor(str(tmp_path / f"{m}.mdfw")))
    got = pr.predict_batch(preds, seqs, coords, q_alns, t_alns)
    assert [g.shape for g in got] == [(10_000, 489), (10_000, 320)] and all(g.dtype == np.float32 for g in got)
    eng = HotPathEngine({m: CtypesPredictor(m, weights=w) for m, w in ws.items()}, device=0, max_rows=65536)
    ref = eng.run_alignments(PackedProteins.pack(seqs, coords, q_alns, t_alns, max_rows=65536))
    assert np.array_equal(got[0], ref["mf"]) and np.array_equal(got[1], ref["cc"])
    for i in (0, 4999, 9999):
        cm = orc.build_align_contact_map(coords[i], q_alns[i], t_alns[i], 6.0, 2)
        assert np.array_equal(got[0][i], preds[0].forward_pass(seqs[i], cm))
        assert np.max(np.abs(got[0][i] - gcn_oracle.gcn_forward(ws["mf"], seqs[i], cm))) < 1e-4
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        pr.predict_batch(preds, ["ACD", "AJD"], [coords[0][:3], coords[1][:3]], ["ACD", "AJD"], ["ACD", "AJD"])


def test_batch_engine_of_the_compiled_binding_pipelines_batches_bitwise(compiled, tmp_path):
    """Round 6: `BatchEngine` of the compiled module -- one engine for many batches, mdf_engine_submit_alignments_host /
    mdf_engine_collect_host behind `run` -- against `predict_batch` (an engine per call, synchronous) on the same lists: the same bits, batch
    after batch, in order; an invalid residue is raised at its batch's collect with the reference's message."""
    _, pr = compiled
    ws = {"mf": synthetic.glorot_gcn_weights(seed=0, n_terms=37), "cc": synthetic.glorot_gcn_weights(seed=2, n_terms=21)}
    preds = []
    for m, w in ws.items():
        weights.save_mdfw(str(tmp_path / f"{m}.mdfw"), w)
        preds.append(pr.Predictor(str(tmp_path / f"{m}.mdfw")))
    sets = [synthetic.synthetic_proteins(seed=400 + k, count=n, length=ln, indel_rate=0.04) for k, (n, ln) in enumerate(((30, (20, 400)), (5, 512), (64, (16, 200))))]
    cols = lambda ps: ([p["seq"] for p in ps], [p["coords"] for p in ps], [p["q_aln"] for p in ps], [p["t_aln"] for p in ps])  # noqa: E731
    refs = [pr.predict_batch(preds, *cols(ps), max_rows=4096) for ps in sets]
    be = pr.BatchEngine(preds, max_rows=4096)
    got = list(be.run([cols(ps) for ps in sets] * 2))
    assert len(got) == 6
    for k, g in enumerate(got):
        assert all(np.array_equal(a, b) for a, b in zip(g, refs[k % 3])), k
    assert all(np.array_equal(a, b) for a, b in zip(be.predict_batch(*cols(sets[1])), refs[1]))
    poisoned = [dict(p) for p in sets[0]]
    k = [i for i, c in enumerate(poisoned[4]["q_aln"]) if c != "-"][1]
    poisoned[4]["seq"] = poisoned[4]["seq"][:1] + "J" + poisoned[4]["seq"][2:]
    poisoned[4]["q_aln"] = poisoned[4]["q_aln"][:k] + "J" + poisoned[4]["q_aln"][k + 1:]
    be.submit(*cols(poisoned))
    be.submit(*cols(sets[2]))
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        be.collect()
    assert all(np.array_equal(a, b) for a, b in zip(be.collect(), refs[2]))


def test_forked_pool_map_build_align_contact_map():
    """reference pipeline.py:476-481 from a parent that has not touched the GPU (a fresh interpreter: this pytest process has)."""
    env = dict(os.environ)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fork_pool_script.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["n"] == 24 and out["none"] == 1 and out["ok"] == 23 and out["names"] == ["q0", "q1", "q2"]
