"""GPU parity of the language-model branch against oracle/lm_oracle.py, through the C ABI.
The oracle for this branch is PARITY UNPINNED (see its header): these tests prove the HIP path computes the restated
Keras arithmetic, not that it matches a released .onnx file."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["persistent", "gemm"], autouse=True)
def lstm_form(request, monkeypatch):
    """Every test runs twice: small groups through the one-launch persistent LSTM kernel (default for <= 16 proteins) and
    through the per-time-step MFMA GEMM form used for large groups (forced with the developer knob)."""
    if request.param == "gemm":
        monkeypatch.setenv("MDFRI_LM_PERSISTENT_MAX_B", "0")
    else:
        monkeypatch.delenv("MDFRI_LM_PERSISTENT_MAX_B", raising=False)
    return request.param


def _weights(seed, hidden, embed, gc, fc, T):
    from mdfri_testkit import synthetic
    w = synthetic.glorot_gcn_weights(seed=seed, n_terms=T, embed=embed, gc_dims=gc, fc_dim=fc)
    w.update(synthetic.glorot_lm_weights(seed=1000, hidden=hidden, embed=embed))
    return w


def _proteins(seed, lengths):
    from mdfri_testkit import synthetic
    rng = np.random.default_rng(seed)
    seqs = [synthetic.random_sequence(rng, L) for L in lengths]
    coords = [synthetic.random_walk_coords(rng, L) for L in lengths]
    return seqs, coords


def _cmap(xyz, thr=6.0):
    d = ((xyz[:, None, :] - xyz[None, :, :])**2).sum(-1)
    return (d < thr * thr).astype(np.int32)


@pytest.mark.parametrize("hidden,lengths", [(64, [1, 7, 33, 70, 70, 2]), (512, [40, 90, 13])])
def test_lstm_features_match_oracle(hidden, lengths):
    import lm_oracle
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    w = _weights(3, hidden, 256, (256,), 256, 20)
    seqs, _ = _proteins(11, lengths)
    eng = HotPathEngine({"mf": Predictor("synthetic", weights=w)}, max_rows=128)   # several chunks, one LSTM batch
    got = eng.lm_features(PackedProteins.pack(seqs, max_rows=128))
    assert len(got) == len(seqs)
    for s, g in zip(seqs, got):
        ref = lm_oracle.lm_forward(w, s)
        assert g.shape == ref.shape
        np.testing.assert_allclose(g, ref, atol=2e-5, rtol=0)


def test_lstm_batches_are_independent():
    """A protein's features do not depend on what it is batched with, nor on the LSTM batch boundaries."""
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    w = _weights(3, 64, 256, (256,), 256, 20)
    seqs, _ = _proteins(5, [50, 20, 64, 31, 8, 64, 3])
    pred = Predictor("synthetic", weights=w)
    a = HotPathEngine({"mf": pred}, max_rows=128).lm_features(PackedProteins.pack(seqs, max_rows=128))
    b = HotPathEngine({"mf": pred}, max_rows=128, lm_batch=2).lm_features(PackedProteins.pack(seqs, max_rows=128))
    for k, s in enumerate(seqs):
        alone = HotPathEngine({"mf": pred}).lm_features(PackedProteins.pack([s]))[0]
        np.testing.assert_array_equal(a[k], alone)
        np.testing.assert_array_equal(b[k], alone)


def test_forward_pass_with_language_model():
    """Per-call API (Predictor.forward_pass, reference predict.pyx:75-102) on a model with the LM branch."""
    import lm_oracle
    from mDeepFRI.predict import Predictor
    w = _weights(7, 64, 256, (256, 256), 256, 33)
    seqs, coords = _proteins(2, [57, 130])
    pred = Predictor("synthetic", weights=w)
    for s, c in zip(seqs, coords):
        A = _cmap(c)
        ref = lm_oracle.gcn_lm_forward(w, s, A)
        got = pred.forward_pass(s, A)
        assert np.abs(got - ref).max() < 1e-4        # north_star tolerance on GO scores


def test_engine_scores_with_language_model_full_size():
    """Batched fused path, shipped topology (LSTM 2x512 -> 1024 embedding -> 512-512-512 -> 1024), two heads sharing
    one language model, mixed with a head that has none."""
    import cmap_oracle
    import gcn_oracle
    import lm_oracle
    from mdfri_testkit import synthetic
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    lm = synthetic.glorot_lm_weights(seed=1000, hidden=512, embed=1024)
    heads = {}
    for k, (mode, T) in enumerate((("mf", 489), ("cc", 320))):
        w = synthetic.glorot_gcn_weights(seed=20 + k, n_terms=T)
        w.update(lm)
        w["W_lm"] = synthetic.glorot_uniform(np.random.default_rng(50 + k), 512, 1024)   # LM_embedding is per head
        heads[mode] = w
    heads["plain"] = synthetic.glorot_gcn_weights(seed=30, n_terms=77)
    preds = {m: Predictor("synthetic", weights=w) for m, w in heads.items()}
    assert preds["mf"].session.lm is preds["cc"].session.lm and preds["plain"].session.lm is None
    prot = synthetic.synthetic_proteins(seed=4, count=6, length=(33, 200), indel_rate=0.05)
    eng = HotPathEngine(preds, max_rows=256)
    pk = PackedProteins.pack([d["seq"] for d in prot], [d["coords"] for d in prot], [d["q_aln"] for d in prot],
                             [d["t_aln"] for d in prot], max_rows=256)
    out = eng.run_alignments(pk)
    for p, d in enumerate(prot):
        A = cmap_oracle.build_align_contact_map(d["coords"], d["q_aln"], d["t_aln"], 6.0, 2)
        for m, w in heads.items():
            ref = lm_oracle.gcn_lm_forward(w, d["seq"], A) if "W_lm" in w else gcn_oracle.gcn_forward(w, d["seq"], A)
            assert np.abs(out[m][p] - ref).max() < 1e-4, (m, p)
    # the same batches through the library's two-slot host pipeline (round 6; the LSTM groups run on the pipeline's compute stream): the same bits
    from mDeepFRI.batch import HostPipeline
    cols = ([d["seq"] for d in prot], [d["coords"] for d in prot], [d["q_aln"] for d in prot], [d["t_aln"] for d in prot])
    got = list(HostPipeline(eng).run([cols, tuple(c[::-1] for c in cols), cols]))
    for m in heads:
        assert np.array_equal(got[0][m], out[m]) and np.array_equal(got[2][m], out[m]) and np.array_equal(got[1][m], out[m][::-1]), m


def test_language_model_head_with_linear_embedding_and_aa_bias():
    """Topology variants of the embedding in a head WITH the language model: no activation on the sum (`embed_linear`), bias on the
    AA branch (`b_aa`) -- vs the oracle, per call and batched."""
    import cmap_oracle
    import lm_oracle
    from mdfri_testkit import synthetic
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    w = synthetic.glorot_gcn_weights(seed=31, n_terms=41, embed=256, gc_dims=(256, 256), fc_dim=256, embed_linear=True, embed_bias=True)
    w.update(synthetic.glorot_lm_weights(seed=1000, hidden=64, embed=256))
    pred = Predictor("synthetic", weights=w)
    prot = synthetic.synthetic_proteins(seed=5, count=5, length=(20, 150), indel_rate=0.05)
    out = HotPathEngine({"m": pred}, max_rows=256).run_alignments(
        PackedProteins.pack([d["seq"] for d in prot], [d["coords"] for d in prot], [d["q_aln"] for d in prot], [d["t_aln"] for d in prot], max_rows=256))["m"]
    for p, d in enumerate(prot):
        A = cmap_oracle.build_align_contact_map(d["coords"], d["q_aln"], d["t_aln"], 6.0, 2)
        ref = lm_oracle.gcn_lm_forward(w, d["seq"], A)
        assert np.abs(out[p] - ref).max() < 1e-4, p
        assert np.abs(pred.forward_pass(d["seq"], A) - ref).max() < 1e-4, p
    plain = {k: v for k, v in w.items() if k not in ("embed_linear", "b_aa")}
    assert np.abs(lm_oracle.gcn_lm_forward(plain, prot[0]["seq"], cmap_oracle.build_align_contact_map(prot[0]["coords"], prot[0]["q_aln"], prot[0]["t_aln"], 6.0, 2)) - out[0]).max() > 1e-5


def test_mdfw_roundtrip_with_language_model(tmp_path):
    from mDeepFRI import weights as W
    from mDeepFRI.predict import Predictor
    w = _weights(9, 64, 256, (256,), 256, 12)
    path = str(tmp_path / "lm_model.mdfw")
    W.save_mdfw(path, w)
    seqs, coords = _proteins(8, [45])
    A = _cmap(coords[0])
    a = Predictor(path).forward_pass(seqs[0], A)
    b = Predictor("synthetic", weights=w).forward_pass(seqs[0], A)
    np.testing.assert_array_equal(a, b)


def test_predictor_loads_an_onnx_file(tmp_path):
    """The path the reference's pipeline passes (an .onnx file, pipeline.py:549-584) is read directly by mDeepFRI.onnx_reader."""
    from mdfri_testkit import onnx_writer
    from mDeepFRI.predict import Predictor
    w = _weights(13, 64, 256, (256, 256), 256, 21)
    path = tmp_path / "DeepFRI-MERGED_GraphConv_gcd_512-512-512_fcd_1024_ca_10.0_mf.onnx"
    path.write_bytes(onnx_writer.deepfri_gcn_model(w))
    seqs, coords = _proteins(8, [61])
    A = _cmap(coords[0])
    a = Predictor(str(path)).forward_pass(seqs[0], A)
    b = Predictor("synthetic", weights=w).forward_pass(seqs[0], A)
    np.testing.assert_array_equal(a, b)


def test_lstm_forms_agree(monkeypatch):
    """The per-time-step GEMM form and the one-launch persistent form compute the same features (different summation order,
    hence a tolerance), for a group and for single proteins."""
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    w = _weights(3, 64, 256, (256,), 256, 20)
    seqs, _ = _proteins(17, [30 + 3 * i for i in range(20)])
    pred = Predictor("synthetic", weights=w)
    monkeypatch.setenv("MDFRI_LM_PERSISTENT_MAX_B", "0")
    gemm = HotPathEngine({"mf": pred}).lm_features(PackedProteins.pack(seqs))
    monkeypatch.delenv("MDFRI_LM_PERSISTENT_MAX_B", raising=False)
    pers = HotPathEngine({"mf": pred}).lm_features(PackedProteins.pack(seqs))
    for k in range(20):
        np.testing.assert_allclose(gemm[k], pers[k], atol=2e-5, rtol=0)
    for k in (0, 7, 19):   # the persistent form is batch-invariant bit for bit
        np.testing.assert_array_equal(pers[k], HotPathEngine({"mf": pred}).lm_features(PackedProteins.pack([seqs[k]]))[0])


def test_heads_sharing_one_language_model_from_two_threads():
    """Every GO head whose file carries the same LM weights shares ONE device-side mdf_lm (sync words, second stream, event ring):
    calls from two threads are serialised inside the library and return the single-threaded results."""
    import threading
    import lm_oracle
    from mDeepFRI.predict import Predictor
    w1, w2 = _weights(3, 64, 256, (256,), 256, 20), _weights(4, 64, 256, (256,), 256, 12)
    p1, p2 = Predictor("lm-a", weights=w1), Predictor("lm-b", weights=w2)
    assert p1.session.lm is p2.session.lm
    seqs, coords = _proteins(5, [60, 35, 90])
    cms = [_cmap(c) for c in coords]
    expect = [[p.forward_pass(s, c) for s, c in zip(seqs, cms)] for p in (p1, p2)]
    assert np.max(np.abs(expect[0][0] - lm_oracle.gcn_lm_forward(w1, seqs[0], cms[0]))) < 1e-4
    bad = []

    def worker(which):
        p = (p1, p2)[which]
        for rep in range(8):
            for i, (s, c) in enumerate(zip(seqs, cms)):
                if not np.array_equal(p.forward_pass(s, c), expect[which][i]):
                    bad.append((which, rep, i))

    ts = [threading.Thread(target=worker, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not bad, bad[:5]
