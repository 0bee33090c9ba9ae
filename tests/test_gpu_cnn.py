"""GPU parity of the sequence-only CNN path (reference predict.pyx:91-100, pipeline.py:600-648)
against oracle/cnn_oracle.py, through the C ABI.  The oracle is PARITY UNPINNED (see its header)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _seqs(seed, lengths):
    from mdfri_testkit import synthetic
    rng = np.random.default_rng(seed)
    return [synthetic.random_sequence(rng, L) for L in lengths]


def test_forward_pass_sequence_only_matches_oracle():
    import cnn_oracle
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    w = synthetic.glorot_cnn_weights(seed=1, n_terms=489)
    pred = Predictor("synthetic-cnn", weights=w)
    assert pred.input_names == ["seq"]                       # predict.pyx:91-95 feeds input_names[0] only
    for s in _seqs(3, [1, 2, 4, 19, 20, 21, 64, 257, 1000]):   # shorter than, equal to and longer than the longest kernel (20)
        got = pred.forward_pass(s)
        ref = cnn_oracle.cnn_forward(w, s)
        assert got.shape == ref.shape == (489,)
        assert np.abs(got - ref).max() < 1e-4, len(s)
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        pred.forward_pass("ACDJ")
    with pytest.raises(ValueError, match="takes no contact map"):
        pred.forward_pass("ACD", np.eye(3, dtype=np.int32))


def test_gcn_model_without_cmap_is_refused():
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    pred = Predictor("synthetic", weights=synthetic.glorot_gcn_weights(seed=0, n_terms=5, embed=64, gc_dims=(256,), fc_dim=256))
    with pytest.raises(ValueError, match="pass the contact map"):
        pred.forward_pass("ACDEFG")


@pytest.mark.parametrize("filters,kernel_lens", [((120, 100, 80, 60), (5, 10, 15, 20)), ((64, 130), (1, 128)), ((512,) * 3, (8, 16, 24))])
def test_sequence_engine_matches_oracle(filters, kernel_lens):
    import cnn_oracle
    from mdfri_testkit import synthetic
    from mDeepFRI.batch import SequenceEngine
    from mDeepFRI.predict import Predictor
    heads = {"mf": synthetic.glorot_cnn_weights(seed=5, n_terms=77, filters=filters, kernel_lens=kernel_lens),
             "cc": synthetic.glorot_cnn_weights(seed=6, n_terms=31, filters=filters, kernel_lens=kernel_lens)}
    eng = SequenceEngine({m: Predictor("synthetic-cnn", weights=w) for m, w in heads.items()}, max_rows=512)   # several chunks
    seqs = _seqs(9, [33, 200, 7, 129, 64, 300, 1, 90])
    out = eng.run(seqs)
    for m, w in heads.items():
        assert out[m].shape == (len(seqs), w["b_out"].shape[0] // 2)
        for i, s in enumerate(seqs):
            assert np.abs(out[m][i] - cnn_oracle.cnn_forward(w, s)).max() < 1e-4, (m, i)


def test_cnn_container_round_trip(tmp_path):
    from mDeepFRI import weights as W
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    w = synthetic.glorot_cnn_weights(seed=2, n_terms=12)
    w["cnn_pad2"] = np.array([3], dtype=np.float32)          # explicit (non-default) left padding survives the container
    path = str(tmp_path / "DeepCNN-MERGED_mf.mdfw")
    W.save_mdfw(path, w)
    s = _seqs(1, [75])[0]
    a = Predictor(path[:-5] + ".onnx").forward_pass(s)       # the pipeline passes the .onnx name; the .mdfw sibling is found
    b = Predictor("synthetic-cnn", weights=w).forward_pass(s)
    np.testing.assert_array_equal(a, b)
    import cnn_oracle
    assert np.abs(a - cnn_oracle.cnn_forward(w, s)).max() < 1e-4


def test_sequence_engine_flags_invalid_residue():
    from mdfri_testkit import synthetic
    from mDeepFRI.batch import SequenceEngine
    from mDeepFRI.predict import Predictor
    eng = SequenceEngine({"mf": Predictor("synthetic-cnn", weights=synthetic.glorot_cnn_weights(seed=1, n_terms=7))}, max_rows=256)
    seqs = _seqs(2, [40, 300, 12])
    seqs[1] = seqs[1][:100] + "j" + seqs[1][101:]            # lowercase is invalid (reference tests/test_predict.py)
    with pytest.raises(ValueError, match="Invalid character in sequence: j"):
        eng.run(seqs)
