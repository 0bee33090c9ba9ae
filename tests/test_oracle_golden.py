"""Pin the oracle (oracle/cmap_oracle.c, oracle/gcn_oracle.py) against the golden vectors produced by the
compiled reference (tests/golden/make_golden.py) and the reference's own known-answer tests.  CPU only."""
import hashlib
import os

import numpy as np
import pytest

import cmap_oracle as orc
import gcn_oracle
from conftest import GOLDEN, gstr
from mdfri_testkit import synthetic


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_pairwise_kat_reference_test_file(cmap_golden):
    # reference mDeepFRI/tests/test_contact_map_utils.py:15-25 (its allclose is never asserted; we assert it)
    X = cmap_golden["pairwise/kat_seed42/X"]
    D = orc.pairwise_sqeuclidean(X)
    assert np.array_equal(D.view(np.uint32), cmap_golden["pairwise/kat_seed42/D"].view(np.uint32))
    assert np.allclose(D, cmap_golden["pairwise/kat_seed42/D_testfile"])


def test_pairwise_collinear_kat(cmap_golden):
    # reference mDeepFRI/tests/test_conctact_map.py:36-41
    X = cmap_golden["pairwise/collinear/X"]
    D = orc.pairwise_sqeuclidean(X)
    assert np.array_equal(D, cmap_golden["pairwise/collinear/D"])
    assert np.array_equal(orc.contacts_lt(D, 6.0), np.array([[1, 1, 0], [1, 1, 1], [0, 1, 1]], dtype=np.int32))


def test_pairwise_generic_shapes_bit_exact(cmap_golden):
    keys = [k[:-2] for k in cmap_golden.files if k.startswith("pairwise/gen_") and k.endswith("/X")]
    assert len(keys) >= 6
    for k in keys:
        X = cmap_golden[k + "/X"]
        D = orc.pairwise_sqeuclidean(X)
        assert D.shape == cmap_golden[k + "/D"].shape
        assert np.array_equal(D.view(np.uint32), cmap_golden[k + "/D"].view(np.uint32)), k


def test_align_cases_bit_exact(cmap_golden):
    names = [str(n) for n in cmap_golden["index/align"]]
    assert len(names) >= 20
    for n in names:
        out = orc.align_contact_map(gstr(cmap_golden[n + "/q"]), gstr(cmap_golden[n + "/t"]),
                                    cmap_golden[n + "/pairs"], int(cmap_golden[n + "/gen"]))
        exp = cmap_golden[n + "/out"]
        assert out.shape == exp.shape and out.dtype == np.int32, n
        assert np.array_equal(out, exp), n


def test_align_symmetrised_inputs_match_reference_test_expectations(cmap_golden):
    # the expectations written in reference tests/test_contact_map_utils.py:38,60,89-95 hold for symmetric input
    assert np.array_equal(cmap_golden["align/identity_sym/out"], np.ones((2, 2), np.int32))
    assert np.array_equal(cmap_golden["align/gap_in_query_sym/out"], np.ones((2, 2), np.int32))
    assert np.array_equal(cmap_golden["align/gap_in_target_sym/out"], np.ones((3, 3), np.int32))
    # and the one-directional inputs give the one-directional .pyx result
    assert np.array_equal(cmap_golden["align/identity_onedir/out"], np.array([[1, 1], [0, 1]], np.int32))


def test_chain_cases_bit_exact(cmap_golden):
    names = [str(n) for n in cmap_golden["index/chain"]]
    assert len(names) >= 24
    for n in names:
        coords = cmap_golden[n + "/coords"]
        q, t, gen = gstr(cmap_golden[n + "/q"]), gstr(cmap_golden[n + "/t"]), int(cmap_golden[n + "/gen"])
        D = orc.pairwise_sqeuclidean(coords)
        assert sha(D) == gstr(cmap_golden[n + "/sha_D"]), n
        sparse = orc.calculate_contact_map(coords, 6.0, mode="sparse")
        assert sparse.shape[0] == int(cmap_golden[n + "/nnz_target"]), n
        assert sha(sparse) == gstr(cmap_golden[n + "/sha_sparse"]), n
        out = orc.align_contact_map(q, t, sparse, gen)
        assert sha(out) == gstr(cmap_golden[n + "/sha_out"]), n
        fused = orc.build_align_contact_map(coords, q, t, 6.0, gen)
        assert np.array_equal(fused, out), n
        if n + "/out" in cmap_golden.files:
            assert np.array_equal(out, cmap_golden[n + "/out"]), n


def test_chain_cases_at_other_thresholds_bit_exact():
    """The same chain at (10 A, 2) -- the operating point of the released `..._ca_10.0_...` GCN files -- (4, 0), (8, 5), (10, 0),
    (7.5, 1): goldens made by the compiled reference (tests/golden/make_thr_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "cmap_thr_golden.npz"))
    names = [str(n) for n in z["index"]]
    assert len(names) == 20
    for n in names:
        coords, q, t = z[n + "/coords"], gstr(z[n + "/q"]), gstr(z[n + "/t"])
        thr, gen = float(z[n + "/thr"]), int(z[n + "/gen"])
        sparse = orc.calculate_contact_map(coords, thr, mode="sparse")
        assert sparse.shape[0] == int(z[n + "/nnz_target"]) and sha(sparse) == gstr(z[n + "/sha_sparse"]), n
        out = orc.align_contact_map(q, t, sparse, gen)
        assert sha(out) == gstr(z[n + "/sha_out"]), n
        assert np.array_equal(orc.build_align_contact_map(coords, q, t, thr, gen), out), n
        if n + "/out_bits" in z.files:
            assert np.array_equal(np.packbits(out.astype(np.uint8), axis=1), z[n + "/out_bits"]), n


def test_special_values_through_the_oracle(cmap_special_golden):
    """the C restatement against the compiled reference on squared distances that are subnormal, overflow to inf, involve
    -0.0 / NaN / inf coordinates or sit exactly on the threshold, at thresholds one float32 step either side of 6 A, tiny, huge (its square
    overflows float32: numpy compares with inf) and zero: D bit for bit, the int32 map per threshold, the aligned map."""
    from conftest import same_float_bits, special_cases
    seen = 0
    for name, X, D_bits, maps, aligned in special_cases(cmap_special_golden):
        D = orc.pairwise_sqeuclidean(X)
        assert same_float_bits(D, D_bits), name
        for thr, cm in maps:
            assert np.array_equal(orc.calculate_contact_map(X, thr), cm), (name, thr)
        seq = "A" * X.shape[0]
        for gen, want in aligned.items():
            assert np.array_equal(orc.build_align_contact_map(X, seq, seq, 6.0, gen), want), (name, gen)
        Dw = np.asarray(D_bits, dtype=np.uint32).view(np.float32)
        seen += int(((Dw > 0) & (Dw < np.finfo(np.float32).tiny)).sum()) + int(np.isinf(Dw).sum()) + int(np.isnan(Dw).sum())
    assert seen > 2500      # the fixtures do hold subnormal, inf and NaN cells


def test_oracle_against_live_reference_when_present():
    """When oracle/_ref (the compiled reference) is available, fuzz the restatement against it directly."""
    import build_ref
    ref = build_ref.load()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(99)
    for it in range(60):
        L = int(rng.integers(1, 90))
        seq = synthetic.random_sequence(rng, L)
        q, t, lt = synthetic.mutate_alignment(rng, seq, float(rng.choice([0.0, 0.1, 0.3])))
        coords = synthetic.random_walk_coords(rng, lt).reshape(-1, 3)
        D = ref.pairwise_sqeuclidean(coords)
        assert np.array_equal(orc.pairwise_sqeuclidean(coords).view(np.uint32), D.view(np.uint32))
        # arbitrary (unsorted, one-directional, partly out-of-range) contact lists
        n_pairs = int(rng.integers(0, 4 * L + 1))
        pairs = rng.integers(-2, lt + 3, size=(n_pairs, 2)).astype(np.int32)
        gen = int(rng.integers(0, 6))
        assert np.array_equal(orc.align_contact_map(q, t, pairs, gen), ref.align_contact_map(q, t, pairs, gen))


# ---- seq2onehot: reference KATs, mDeepFRI/tests/test_predict.py:9-33
def test_seq2onehot_reference_kats():
    r = orc.seq2onehot("")
    assert r.shape == (0, 26) and r.dtype == np.float32
    assert np.array_equal(orc.seq2onehot("D"), np.array([[0, 1] + [0] * 24], dtype=np.float32))
    exp = np.zeros((4, 26), np.float32)
    exp[np.arange(4), np.arange(4)] = 1
    assert np.array_equal(orc.seq2onehot("-DGU"), exp)
    with pytest.raises(ValueError):
        orc.seq2onehot("J")


def test_seq2onehot_alphabet_and_invalid_classes():
    alpha = "-DGULNTKHYWCPVSOIEFXQABZRM"  # reference predict.pyx:26
    assert np.array_equal(orc.seq2onehot(alpha), np.eye(26, dtype=np.float32))
    assert np.array_equal(gcn_oracle.onehot(alpha), np.eye(26, dtype=np.float32))
    for bad in ("a", "*", "AC DE", "ACj"):
        with pytest.raises(ValueError, match="Invalid character in sequence"):
            orc.seq2onehot(bad)


# ---- GCN restatement: regression anchors (parity unpinned; see oracle/gcn_oracle.py header)
def test_gcn_oracle_regression_and_fp32_vs_fp64(gcn_golden):
    names = [str(n) for n in gcn_golden["index/gcn"]]
    assert len(names) == 8
    wcache = {}
    for n in names:
        seq = gstr(gcn_golden[n + "/seq"])
        L = len(seq)
        cm = np.unpackbits(gcn_golden[n + "/cmap_bits"], axis=1)[:, :L].astype(np.int32)
        key = (int(gcn_golden[n + "/wseed"]), int(gcn_golden[n + "/n_terms"]))
        if key not in wcache:
            wcache[key] = synthetic.glorot_gcn_weights(seed=key[0], n_terms=key[1])
        y32 = gcn_oracle.gcn_forward(wcache[key], seq, cm, dtype=np.float32)
        y64 = gcn_golden[n + "/y64"]
        assert y32.shape == (key[1],) and y32.dtype == np.float32
        # fp32 op order of the reference graph vs float64 ground truth: far inside the 1e-4 budget
        assert np.max(np.abs(y32.astype(np.float64) - y64)) < 2e-5, n
        # scores must be informative (not saturated), else a 1e-4 absolute check would be vacuous
        assert np.mean((y64 > 0.02) & (y64 < 0.98)) > 0.5, n


def test_gcn_oracle_normalisation_properties():
    rng = np.random.default_rng(3)
    A = rng.integers(0, 2, size=(17, 17)).astype(np.float32)
    Ah = gcn_oracle.normalize_adjacency(A)
    # diagonal of A is replaced by 1 whatever it was (A - diag(A) + I)
    A2 = A.copy()
    np.fill_diagonal(A2, 7.0)
    assert np.array_equal(gcn_oracle.normalize_adjacency(A2), Ah)
    d = 1.0 / (1e-6 + np.sqrt((A - np.diag(np.diag(A)) + np.eye(17)).sum(1)))
    assert np.allclose(Ah, d[:, None] * (A - np.diag(np.diag(A)) + np.eye(17)) * d[None, :], rtol=1e-6)
