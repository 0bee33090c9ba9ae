"""GPU parity: DeepFRI GCN forward (HIP) vs the CPU restatement oracle/gcn_oracle.py.
Tolerance: 1e-4 absolute on the scores (BASELINE.json north_star; reference notebook atol 10e-5).
The oracle itself is *parity unpinned* against the reference's ONNX path (see its header)."""
import os

import numpy as np
import pytest

import cmap_oracle as orc
import gcn_oracle
from conftest import gstr
from mdfri_testkit import synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def mf():
    from mDeepFRI.predict import Predictor
    w = synthetic.glorot_gcn_weights(seed=0, n_terms=synthetic.GO_TERMS["mf"])
    return w, Predictor("synthetic-mf.onnx", weights=w)


@pytest.fixture(scope="module")
def cc():
    from mDeepFRI.predict import Predictor
    w = synthetic.glorot_gcn_weights(seed=2, n_terms=synthetic.GO_TERMS["cc"])
    return w, Predictor("synthetic-cc.onnx", weights=w)


def test_predictor_public_surface(mf):
    _, pred = mf
    assert pred.model_path == "synthetic-mf.onnx" and pred.threads == 1
    assert pred.input_names == ["cmap", "seq"] and pred.session is not None
    with pytest.raises(ValueError, match="pass the contact map"):   # a GCN file has two inputs; sequence-only needs a DeepCNN file
        pred.forward_pass("ACD")
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        pred.forward_pass("AJD", np.eye(3, dtype=np.int32))
    with pytest.raises(ValueError):
        pred.forward_pass("ACD", np.eye(4, dtype=np.int32))


def test_invalid_residue_is_the_first_one_per_call(mf):
    """reference predict.pyx:36-46 scans left to right and raises on the FIRST invalid byte; with several offenders the
    device's 64-bit atomic minimum must report that one, whichever thread gets there first."""
    _, pred = mf
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        pred.forward_pass("AJD*Z", np.eye(5, dtype=np.int32))
    seq = "ACDE" * 200 + "*" + "j" * 300 + "J" * 200 + "ACD"
    for _ in range(20):
        with pytest.raises(ValueError, match=r"Invalid character in sequence: \*"):
            pred.forward_pass(seq, np.eye(len(seq), dtype=np.int32))


@pytest.mark.parametrize("max_rows", [128, 65536])
def test_invalid_residue_in_a_batch_is_the_lowest_protein_lowest_position(mf, max_rows):
    """Bad letters in proteins 7 and 3 (one chunk, and spread over several chunks): the batch reports protein 3's first one,
    as the reference's serial loop over the proteins would."""
    from mDeepFRI.batch import PackedProteins
    _, pred = mf
    prots = synthetic.synthetic_proteins(seed=8, count=12, length=(60, 120))
    seqs = [p["seq"] for p in prots]
    seqs[7] = "J" + seqs[7][1:]
    s3 = list(seqs[3])
    s3[40], s3[2], s3[17] = "x", "*", "j"
    seqs[3] = "".join(s3)
    eng = _engine({"mf": pred}, max_rows=max_rows)
    pk = PackedProteins.pack(seqs, [p["coords"] for p in prots], seqs, seqs, max_rows=max_rows)
    assert (len(pk.chunks) > 4) == (max_rows == 128)
    for _ in range(5):
        with pytest.raises(ValueError, match=r"Invalid character in sequence: \*"):
            eng.run_alignments(pk)


def test_forward_pass_golden_cases(gcn_golden, mf, cc):
    for n in [str(x) for x in gcn_golden["index/gcn"]]:
        w, pred = mf if "/mf_" in n else cc
        seq = gstr(gcn_golden[n + "/seq"])
        L = len(seq)
        cm = np.unpackbits(gcn_golden[n + "/cmap_bits"], axis=1)[:, :L].astype(np.int32)
        y = pred.forward_pass(seq, cm)
        assert y.dtype == np.float32 and y.shape == (pred.n_terms,)
        err64 = np.max(np.abs(y.astype(np.float64) - gcn_golden[n + "/y64"]))
        err32 = np.max(np.abs(y - gcn_oracle.gcn_forward(w, seq, cm)))
        assert err64 < TOL and err32 < TOL, (n, err64, err32)


@pytest.mark.parametrize("dtype", [np.int32, np.float32, np.int64, np.float64, np.uint8, np.bool_, np.int16])
def test_forward_pass_accepts_cmap_dtypes(mf, dtype):
    w, pred = mf
    rng = np.random.default_rng(1)
    seq = synthetic.random_sequence(rng, 90)
    cm = orc.calculate_contact_map(synthetic.random_walk_coords(rng, 90), 6.0)
    y = pred.forward_pass(seq, cm.astype(dtype))
    assert np.max(np.abs(y - gcn_oracle.gcn_forward(w, seq, cm))) < TOL


def test_forward_pass_general_float_and_asymmetric_maps(mf):
    """forward_pass takes any (L,L) numeric matrix (reference docstring: 'binary or distance-based'): weighted,
    non-symmetric maps with a non-unit diagonal go through the same normalisation as the oracle."""
    w, pred = mf
    rng = np.random.default_rng(77)
    L = 150
    seq = synthetic.random_sequence(rng, L)
    A = (rng.random((L, L)) < 0.1) * rng.random((L, L))
    np.fill_diagonal(A, 3.0)
    A = A.astype(np.float32)
    y = pred.forward_pass(seq, A)
    assert np.max(np.abs(y - gcn_oracle.gcn_forward(w, seq, A))) < TOL


@pytest.mark.parametrize("L", [1, 2, 31, 32, 33, 127, 128, 129, 512, 1024])
def test_forward_pass_lengths(mf, L):
    w, pred = mf
    rng = np.random.default_rng(100 + L)
    seq = synthetic.random_sequence(rng, L)
    cm = orc.calculate_contact_map(synthetic.random_walk_coords(rng, L), 6.0)
    y = pred.forward_pass(seq, cm)
    ref = gcn_oracle.gcn_forward(w, seq, cm)
    assert np.max(np.abs(y - ref)) < TOL, L
    assert np.mean((ref > 0.02) & (ref < 0.98)) > 0.3  # the check is not vacuous (scores are not saturated)


def test_all_26_letters(mf):
    w, pred = mf
    seq = "-DGULNTKHYWCPVSOIEFXQABZRM" * 3
    rng = np.random.default_rng(4)
    cm = orc.calculate_contact_map(synthetic.random_walk_coords(rng, len(seq)), 6.0)
    assert np.max(np.abs(pred.forward_pass(seq, cm) - gcn_oracle.gcn_forward(w, seq, cm))) < TOL


def _engine(preds, **kw):
    from mDeepFRI.batch import HotPathEngine
    return HotPathEngine(preds, device=0, **kw)


def test_fused_batch_vs_oracle_and_per_call(mf, cc):
    """coords + alignment + sequence -> scores in one fused batch == oracle chain == per-call drop-in chain."""
    from mDeepFRI.batch import PackedProteins
    (wm, pm), (wc, pc) = mf, cc
    prots = synthetic.synthetic_proteins(seed=21, count=24, length=(30, 300), indel_rate=0.06)
    eng = _engine({"mf": pm, "cc": pc}, max_rows=2048)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots], max_rows=2048)
    assert len(pk.chunks) > 1
    db = eng.upload(pk)
    scores, logits = eng.forward_alignments(db, want_logits=True)
    eng.check(db)
    s_mf, s_cc = scores["mf"].cpu().numpy(), scores["cc"].cpu().numpy()
    z_mf = logits["mf"].cpu().numpy()
    for i, p in enumerate(prots):
        cm = orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)
        y_mf, im = gcn_oracle.gcn_forward(wm, p["seq"], cm, dtype=np.float64, return_intermediates=True)
        assert np.max(np.abs(s_mf[i] - y_mf)) < TOL, i
        assert np.max(np.abs(s_cc[i] - gcn_oracle.gcn_forward(wc, p["seq"], cm))) < TOL, i
        # pre-softmax logits: tighter, saturation-proof check
        z64 = (im["f"] @ wm["W_out"].astype(np.float64) + wm["b_out"]).reshape(-1)
        assert np.max(np.abs(z_mf[i] - z64)) < 2e-3 * max(1.0, np.abs(z64).max()), i
        # the per-call drop-in runs the same kernels in the same summation order: identical, not merely close
        assert np.array_equal(s_mf[i], pm.forward_pass(p["seq"], cm)), i


@pytest.fixture(scope="module")
def bp():
    from mDeepFRI.predict import Predictor
    w = synthetic.glorot_gcn_weights(seed=1, n_terms=synthetic.GO_TERMS["bp"])
    return w, Predictor("synthetic-bp.onnx", weights=w)


@pytest.mark.parametrize("threshold,gen", [(10.0, 2), (4.0, 0), (8.0, 5)])
def test_fused_batch_at_other_thresholds_and_generated_contacts(mf, threshold, gen):
    """The released GCN files are trained on 10 A maps (`..._ca_10.0_...`, reference mDeepFRI/__init__.py:73,78) while the CLI's
    default is 6 A / generated_contacts 2 (cli.py:360-371): the fused batched path (k_cmap_rows<COUNT> + k_cmap_fill -> CSR ->
    GraphConv) at three other (threshold, generated_contacts) settings vs the oracle chain, with indels so that synthetic
    contacts exist, several chunks, and the CSR capacity retry (10 A gives ~40 entries per row; nnz_per_row starts at 8)."""
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    w, pred = mf
    prots = synthetic.synthetic_proteins(seed=int(threshold) * 10 + gen, count=18, length=(20, 330), indel_rate=0.12)
    eng = HotPathEngine({"mf": pred}, device=0, max_rows=1024, nnz_per_row=8, threshold=threshold, generated_contacts=gen)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots], max_rows=1024)
    assert len(pk.chunks) > 2
    out = eng.run_alignments(pk)["mf"]
    n_syn = 0
    for i, p in enumerate(prots):
        cm = orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], threshold, gen)
        n_syn += int(cm.sum() - orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], threshold, 0).sum())
        assert np.max(np.abs(out[i] - gcn_oracle.gcn_forward(w, p["seq"], cm))) < TOL, (i, len(p["seq"]))
        assert np.array_equal(out[i], pred.forward_pass(p["seq"], cm)), i            # batch == per call, bitwise
    assert (n_syn > 0) == (gen > 0)


@pytest.mark.parametrize("variant", ["embed_linear", "embed_bias", "linear_and_bias", "no_embedding"])
def test_embedding_topology_variants_vs_oracle(variant, tmp_path):
    """The embedding's activation / bias / existence is data the model file decides (include/mdfri.h `embed_linear`, weights `b_aa`,
    identity embedding for a graph without one): per call, batched, through a .mdfw container loaded by the C loader, and through an
    exported .onnx read back -- all vs the oracle, batch == per call bitwise."""
    import ctypes
    from mDeepFRI import _hip, weights
    from mdfri_testkit import onnx_writer
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    kw = {"embed_linear": dict(embed_linear=True), "embed_bias": dict(embed_bias=True), "linear_and_bias": dict(embed_linear=True, embed_bias=True),
          "no_embedding": {}}[variant]
    w = synthetic.glorot_gcn_weights(seed=12, n_terms=57, **kw)
    if variant == "no_embedding":
        del w["W_aa"]
        w["W_gc1"] = synthetic.glorot_uniform(np.random.default_rng(3), 26, 512)
        path = tmp_path / "noembed_mf.onnx"
        path.write_bytes(onnx_writer.deepfri_gcn_model(w))
        pred = Predictor(str(path))                       # the reader expresses it as an identity embedding without activation
        w = dict(w, W_aa=np.eye(26, dtype=np.float32), embed_linear=np.ones(1, np.float32))
    else:
        path = tmp_path / "variant_mf.onnx"
        path.write_bytes(onnx_writer.deepfri_gcn_model(w))
        pred = Predictor(str(path))
        assert weights.validate(weights.load_weights(str(path)))["embed_linear"] == ("linear" in variant)
    prots = synthetic.synthetic_proteins(seed=70, count=9, length=(15, 300), indel_rate=0.05)
    out = HotPathEngine({"m": pred}, device=0, max_rows=1024).run_alignments(
        PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots], max_rows=1024))["m"]
    for i, p in enumerate(prots):
        cm = orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)
        ref = gcn_oracle.gcn_forward(w, p["seq"], cm)
        assert np.max(np.abs(out[i] - ref)) < TOL, (variant, i)
        assert np.array_equal(out[i], pred.forward_pass(p["seq"], cm)), (variant, i)
    # the flag matters: the default topology scores differently on the same tensors
    plain = {k: v for k, v in w.items() if k not in ("embed_linear", "b_aa")}
    cm0 = orc.build_align_contact_map(prots[0]["coords"], prots[0]["q_aln"], prots[0]["t_aln"], 6.0, 2)
    if variant != "no_embedding":      # (an identity embedding is non-negative: relu changes nothing there)
        assert np.max(np.abs(gcn_oracle.gcn_forward(plain, prots[0]["seq"], cm0) - out[0])) > 1e-5
    # the C loader (mdf_model_load) reads the same container
    weights.save_mdfw(str(tmp_path / "v.mdfw"), w)
    h = ctypes.c_void_p()
    _hip.check(_hip.lib().mdf_model_load(str(tmp_path / "v.mdfw").encode(), 0, ctypes.byref(h)))
    try:
        y = np.empty(57, dtype=np.float32)
        bad = _hip.c_int64(-1)
        seq = prots[0]["seq"].encode()
        _hip.check(_hip.lib().mdf_gcn_forward_host(h, seq, len(seq), _hip.ptr(np.ascontiguousarray(cm0)), _hip.DT_I32, _hip.ptr(y), bad))
        assert np.array_equal(y, out[0])
    finally:
        _hip.lib().mdf_model_free(h)


@pytest.mark.parametrize("L", [1, 33, 128, 512, 1024])
def test_bp_sized_head_vs_oracle(bp, L):
    """The biological-process head: T = 1 943 terms -> 3 886 output columns, padded to 4 096 inside the library (the widest GO head
    of the v1.0 models, reference mDeepFRI/__init__.py:47-80); per call and inside a fused batch, both vs the oracle."""
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    w, pred = bp
    assert pred.n_terms == 1943
    p = synthetic.synthetic_proteins(seed=900 + L, count=1, length=L, indel_rate=0.05 if L > 8 else 0.0)[0]
    cm = orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)
    ref = gcn_oracle.gcn_forward(w, p["seq"], cm)
    y = pred.forward_pass(p["seq"], cm)
    assert y.shape == (1943,) and y.dtype == np.float32
    assert np.max(np.abs(y - ref)) < TOL
    eng = HotPathEngine({"bp": pred}, device=0, max_rows=4096)
    others = synthetic.synthetic_proteins(seed=901 + L, count=5, length=(10, 200))
    prots = others[:2] + [p] + others[2:]
    pk = PackedProteins.pack([q["seq"] for q in prots], [q["coords"] for q in prots], [q["q_aln"] for q in prots], [q["t_aln"] for q in prots],
                             max_rows=4096)
    out = eng.run_alignments(pk)["bp"]
    assert np.array_equal(out[2], y)                                                    # company-invariant, bitwise
    for i, q in enumerate(prots):
        cmq = orc.build_align_contact_map(q["coords"], q["q_aln"], q["t_aln"], 6.0, 2)
        assert np.max(np.abs(out[i] - gcn_oracle.gcn_forward(w, q["seq"], cmq))) < TOL, i


@pytest.mark.parametrize("L", [17, 300, 777])
def test_ec_sized_head_vs_oracle(L):
    """The enzyme-commission head of the v1.0 models (reference mDeepFRI/__init__.py:73-80, mode `ec`): T = 538 terms -> 1 076 output
    columns, padded to 1 280 inside the library.  T is data, so the path is the one the GO heads take; this runs it once at that size:
    per call and inside a fused batch of mixed lengths (16-row groups: proteins start in the middle of an MFMA tile), vs the oracle."""
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    w = synthetic.glorot_gcn_weights(seed=4, n_terms=synthetic.GO_TERMS["ec"])
    pred = Predictor("synthetic-ec.onnx", weights=w)
    assert pred.n_terms == 538
    p = synthetic.synthetic_proteins(seed=950 + L, count=1, length=L, indel_rate=0.05)[0]
    cm = orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)
    y = pred.forward_pass(p["seq"], cm)
    assert y.shape == (538,) and np.max(np.abs(y - gcn_oracle.gcn_forward(w, p["seq"], cm))) < TOL
    others = synthetic.synthetic_proteins(seed=951 + L, count=6, length=(5, 150))
    prots = others[:3] + [p] + others[3:]
    pk = PackedProteins.pack([q["seq"] for q in prots], [q["coords"] for q in prots], [q["q_aln"] for q in prots], [q["t_aln"] for q in prots],
                             max_rows=2048)
    out = HotPathEngine({"ec": pred}, device=0, max_rows=2048).run_alignments(pk)["ec"]
    assert np.array_equal(out[3], y)
    for i, q in enumerate(prots):
        cmq = orc.build_align_contact_map(q["coords"], q["q_aln"], q["t_aln"], 6.0, 2)
        assert np.max(np.abs(out[i] - gcn_oracle.gcn_forward(w, q["seq"], cmq))) < TOL, i


def test_fused_batch_on_protein_like_traces(mf):
    """The same chain on helix-bundle traces (~8.6 entries per CSR row instead of ~12.6) with indels: contact
    maps bit-exact with the oracle's, scores within the budget, batch == per call."""
    from mDeepFRI.batch import PackedProteins
    from types import SimpleNamespace
    from mDeepFRI.bio_utils import build_align_contact_map
    w, pred = mf
    prots = synthetic.synthetic_proteins(seed=77, count=20, length=(25, 400), indel_rate=0.08, coords="helix")
    eng = _engine({"mf": pred}, max_rows=2048)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots], max_rows=2048)
    db = eng.upload(pk)
    s = eng.forward_alignments(db)["mf"]
    eng.check(db)
    s = s.cpu().numpy()
    for i, p in enumerate(prots):
        cm = orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)
        assert np.max(np.abs(s[i] - gcn_oracle.gcn_forward(w, p["seq"], cm))) < TOL, i
        assert np.array_equal(s[i], pred.forward_pass(p["seq"], cm)), i
        if i < 6:
            aln = SimpleNamespace(query_name=p["id"], target_name="t", gapped_sequence=p["q_aln"], gapped_target=p["t_aln"], coords=p["coords"])
            assert np.array_equal(build_align_contact_map(aln)[1], cm), i


def test_dense_batch_path_matches_per_call(mf):
    from mDeepFRI.batch import PackedProteins
    w, pred = mf
    prots = synthetic.synthetic_proteins(seed=5, count=9, length=(40, 200))
    cms = [orc.calculate_contact_map(p["coords"], 6.0) for p in prots]
    eng = _engine({"mf": pred}, max_rows=1024)
    pk = PackedProteins.pack([p["seq"] for p in prots], max_rows=1024)
    db = eng.upload(pk)
    s = eng.forward_dense(db, cms)["mf"]
    eng.check(db)
    s = s.cpu().numpy()
    for i, p in enumerate(prots):
        assert np.max(np.abs(s[i] - gcn_oracle.gcn_forward(w, p["seq"], cms[i]))) < TOL, i


def test_batch_flags_invalid_residue(mf):
    from mDeepFRI.batch import PackedProteins
    _, pred = mf
    prots = synthetic.synthetic_proteins(seed=6, count=3, length=50)
    seqs = [p["seq"] for p in prots]
    seqs[1] = seqs[1][:10] + "J" + seqs[1][11:]
    q = list(seqs)
    eng = _engine({"mf": pred})
    pk = PackedProteins.pack(seqs, [p["coords"] for p in prots], q, q)
    db = eng.upload(pk)
    eng.forward_alignments(db)
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        eng.check(db)


def test_csr_overflow_is_detected_and_recovered(mf):
    from mDeepFRI import _hip
    from mDeepFRI.batch import PackedProteins
    w, pred = mf
    prots = synthetic.synthetic_proteins(seed=9, count=4, length=120)
    eng = _engine({"mf": pred}, nnz_per_row=2)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots])
    db = eng.upload(pk)
    eng.forward_alignments(db)
    with pytest.raises(_hip.CapacityError):
        eng.check(db)
    out = eng.run_alignments(pk)["mf"]  # grows the capacity and re-runs
    cm = orc.build_align_contact_map(prots[0]["coords"], prots[0]["q_aln"], prots[0]["t_aln"], 6.0, 2)
    assert np.max(np.abs(out[0] - gcn_oracle.gcn_forward(w, prots[0]["seq"], cm))) < TOL


def test_full_size_batch_properties(mf):
    """BASELINE config shape (L=512 batch): permutation equivariance across the batch and batch-size independence --
    properties that need no oracle at full size -- plus an oracle spot check."""
    from mDeepFRI.batch import PackedProteins
    w, pred = mf
    prots = synthetic.synthetic_proteins(seed=44, count=96, length=512)
    eng = _engine({"mf": pred})

    def run(sel):
        pk = PackedProteins.pack([prots[i]["seq"] for i in sel], [prots[i]["coords"] for i in sel],
                                 [prots[i]["q_aln"] for i in sel], [prots[i]["t_aln"] for i in sel])
        return eng.run_alignments(pk)["mf"]

    full = run(list(range(96)))
    perm = list(np.random.default_rng(0).permutation(96))
    assert np.array_equal(run(perm), full[perm])        # bitwise: no cross-protein coupling, deterministic pooling
    assert np.array_equal(run([7]), full[[7]])
    # ... nor on the GEMM kernel it went through: the 96-protein batch runs the 256x256-tile kernel, a single protein (here and
    # in the per-call API) the one-wave-per-32x32-tile kernel for small problems -- same MFMA k order, identical bits
    for i in (3, 50):
        cm_i = orc.build_align_contact_map(prots[i]["coords"], prots[i]["q_aln"], prots[i]["t_aln"], 6.0, 2)
        assert np.array_equal(pred.forward_pass(prots[i]["seq"], cm_i), full[i])
    for i in (0, 95):
        cm = orc.build_align_contact_map(prots[i]["coords"], prots[i]["q_aln"], prots[i]["t_aln"], 6.0, 2)
        assert np.max(np.abs(full[i] - gcn_oracle.gcn_forward(w, prots[i]["seq"], cm))) < TOL


def test_configs1_workload_1000_proteins_L256_mf(mf):
    """BASELINE.json configs[1]: 1 000 synthetic L=256 proteins, GCN_MF, one GPU.  The oracle checks a sample (it needs
    ~15 ms per protein); ALL 1 000 are covered by size-independent properties: the result of every protein is bitwise
    independent of the order, the chunking and the company it is batched with."""
    from mDeepFRI.batch import PackedProteins
    w, pred = mf
    n, L = 1000, 256
    lengths = np.full(n, L, dtype=np.int32)
    seqs, coords, q_alns, t_alns = synthetic.bulk_proteins(43, lengths, range(n))        # seed = 42 + config index
    eng = _engine({"mf": pred}, max_rows=65536)

    def run(sel, max_rows):
        pk = PackedProteins.pack([seqs[i] for i in sel], [coords[i] for i in sel], [q_alns[i] for i in sel], [t_alns[i] for i in sel],
                                 max_rows=max_rows)
        return eng.run_alignments(pk)["mf"], len(pk.chunks)

    full, n_chunks = run(range(n), 65536)
    assert full.shape == (n, pred.n_terms) and n_chunks == 4 and np.isfinite(full).all()
    for i in np.linspace(0, n - 1, 10).astype(int):
        cm = orc.build_align_contact_map(coords[i], q_alns[i], t_alns[i], 6.0, 2)
        assert np.max(np.abs(full[i] - gcn_oracle.gcn_forward(w, seqs[i], cm))) < TOL, i
    perm = np.random.default_rng(1).permutation(n)
    shuffled, n_chunks2 = run(list(perm), 8192)
    assert n_chunks2 == 32 and np.array_equal(shuffled, full[perm])
    assert np.array_equal(run([123, 7], 65536)[0], full[[123, 7]])
    assert len({row.tobytes() for row in full}) == n        # no two proteins collapse onto the same scores


def test_model_file_round_trip(tmp_path, mf):
    from mDeepFRI import weights
    from mDeepFRI.predict import Predictor
    w, pred = mf
    path = tmp_path / "DeepFRI-SYNTH_GraphConv_gcd_512-512-512_fcd_1024_ca_10.0_mf.mdfw"
    weights.save_mdfw(str(path), w)
    p2 = Predictor(str(path).replace(".mdfw", ".onnx"))  # the pipeline passes the .onnx name; the sibling .mdfw is used
    rng = np.random.default_rng(12)
    seq = synthetic.random_sequence(rng, 64)
    cm = orc.calculate_contact_map(synthetic.random_walk_coords(rng, 64), 6.0)
    assert np.array_equal(p2.forward_pass(seq, cm), pred.forward_pass(seq, cm))


def test_rows_not_multiple_of_gemm_tile_do_not_write_out_of_bounds(mf):
    """R (multiple of 128) need not be a multiple of the 256-row GEMM tile: the half-empty last tile must neither store
    activations nor pool partials past R.  Canary-padded buffers around the workspace and the partial array."""
    import ctypes
    import torch
    from mDeepFRI import _hip
    w, pred = mf
    L = _hip.lib()
    h = pred.session.handle
    R, feat = 384, pred.session.topology["feature_dim"]   # 1.5 tiles
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(0)
    # a trivial adjacency: every row its own neighbour (self loops only), letter sums = one-hot of a fixed letter
    grp = torch.arange(R + 1, dtype=torch.int32, device=dev)
    col = torch.arange(R, dtype=torch.int32, device=dev)
    val = torch.ones(R, dtype=torch.float32, device=dev)
    S = torch.zeros((R, 32), dtype=torch.float32, device=dev)
    S[:, 5] = 1.0
    ws_bytes = L.mdf_gcn_workspace_bytes(h, R)
    pad = 1 << 16
    ws = torch.full((ws_bytes + 2 * pad,), 0x5A, dtype=torch.uint8, device=dev)
    part = torch.full((R // 16 * feat + 2 * pad,), 777.0, dtype=torch.float32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _hip.check(L.mdf_gcn_embed_dev(h, _hip.ptr(S), _hip.ptr(grp), _hip.ptr(col), _hip.ptr(val), R,
                                   ctypes.c_void_p(part.data_ptr() + pad * 4), ctypes.c_void_p(ws.data_ptr() + pad), ws_bytes, st))
    torch.cuda.synchronize()
    assert bool((ws[:pad] == 0x5A).all()) and bool((ws[pad + ws_bytes:] == 0x5A).all())
    assert bool((part[:pad] == 777.0).all()) and bool((part[pad + R // 16 * feat:] == 777.0).all())
    assert bool(torch.isfinite(part[pad:pad + R // 16 * feat]).all())


def _library_first_script(then_torch: bool) -> str:
    from conftest import ROOT
    code = (
        "import sys, os; ROOT = %r; sys.path.insert(0, os.path.join(ROOT, 'metagenomic-deepfri_amd')); sys.path.insert(0, ROOT)\n"
        "import numpy as np\n"
        "from mdfri_testkit import synthetic\n"
        "from mDeepFRI.predict import Predictor\n"
        "p = Predictor('syn', weights=synthetic.glorot_gcn_weights(0, 16))\n"
        "y = p.forward_pass('ACDEFGHIKL', np.eye(10, dtype=np.int32))\n"
        "assert 'torch' not in sys.modules and y.shape == (16,)\n"
        "maps = sorted({ln.split()[-1] for ln in open('/proc/self/maps') if 'libamdhip64' in ln})\n"
        "print('HIPLIBS', maps)\n" % ROOT)
    if then_torch:
        code += (
            "import torch\n"
            "assert torch.cuda.is_available()\n"
            "from mDeepFRI import batch\n"
            "e = batch.HotPathEngine({'mf': p}, device=0)\n"
            "prots = synthetic.synthetic_proteins(1, 3, 50)\n"
            "pk = batch.PackedProteins.pack([q['seq'] for q in prots], [q['coords'] for q in prots], [q['q_aln'] for q in prots], [q['t_aln'] for q in prots])\n"
            "print(e.run_alignments(pk)['mf'].shape)\n")
    return code


def test_library_first_uses_torchs_hip_runtime():
    """Import-order regression: creating a Predictor (libmdfri_hip -> HIP) BEFORE torch is imported used to leave torch.cuda
    unavailable (system ROCm runtime loaded first, torch's bundled one shadowed).  The fix pre-loads torch's bundled runtime
    without importing torch: after the library has computed, exactly ONE libamdhip64 is mapped and it is torch's -- which is what
    lets a later `import torch` share the device state (the full sequence is the slow test below)."""
    import importlib.util
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", _library_first_script(False)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    libs = eval(out.stdout.split("HIPLIBS", 1)[1].strip().splitlines()[0])
    torch_dir = os.path.dirname(importlib.util.find_spec("torch").origin)
    assert len(libs) == 1 and libs[0].startswith(os.path.join(torch_dir, "lib")), libs


@pytest.mark.skipif(not os.environ.get("MDFRI_SLOW_TESTS"), reason="`import torch` after HIP is initialised loads every fat binary of "
                    "libtorch_hip eagerly: 2.5 s warm, 200+ s on a fresh box (tools/import_order_probe.py); set MDFRI_SLOW_TESTS=1")
def test_library_first_then_torch_share_one_hip_runtime():
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", _library_first_script(True)], capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "(3, 16)" in out.stdout


def test_protein_longer_than_a_chunk(mf):
    """A protein longer than max_rows gets an oversized chunk of its own; neighbours of other lengths share the batch."""
    from mDeepFRI.batch import PackedProteins
    w, pred = mf
    prots = synthetic.synthetic_proteins(seed=55, count=1, length=40) + synthetic.synthetic_proteins(seed=56, count=1, length=2500, indel_rate=0.02) \
        + synthetic.synthetic_proteins(seed=57, count=2, length=(100, 200))
    eng = _engine({"mf": pred}, max_rows=1024)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots], max_rows=1024)
    assert max(c.rows for c in pk.chunks) > 1024
    out = eng.run_alignments(pk)["mf"]
    for i, p in enumerate(prots):
        cm = orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)
        assert np.max(np.abs(out[i] - gcn_oracle.gcn_forward(w, p["seq"], cm))) < TOL, i


def test_protein_with_more_than_64_contact_words(mf):
    """L = 4 500 needs 71 contact words per row: the CSR fill spreads a wave's 8 x 71 (row, word) items over several batches of 64
    lanes and must keep every (row, letter) sum in ascending-column order across them -- checked bit for bit against the per-call
    path (dense map -> k_dense_rows + k_letter_sums) and against the oracle within tolerance."""
    from mDeepFRI.batch import PackedProteins
    w, pred = mf
    prots = synthetic.synthetic_proteins(seed=61, count=1, length=4500, indel_rate=0.01) + synthetic.synthetic_proteins(seed=62, count=2, length=(70, 90))
    eng = _engine({"mf": pred}, max_rows=2048)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots], max_rows=2048)
    out = eng.run_alignments(pk)["mf"]
    for i, p in enumerate(prots):
        cm = orc.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2)
        assert np.array_equal(out[i], pred.forward_pass(p["seq"], cm)), i
        assert np.max(np.abs(out[i] - gcn_oracle.gcn_forward(w, p["seq"], cm))) < TOL, i


def test_alignment_stream_matches_one_batch(mf, cc):
    """Host-in / host-out streaming runner: several device batches, a producer thread packing ahead, results in input order
    and bitwise equal to one big batch; objects with the AlignmentResult attributes are accepted as well as tuples."""
    from mDeepFRI.batch import PackedProteins
    from mDeepFRI.stream import AlignmentStream
    (wm, pm), (wc, pc) = mf, cc
    prots = synthetic.synthetic_proteins(seed=31, count=53, length=(20, 260), indel_rate=0.05)
    eng = _engine({"mf": pm, "cc": pc}, max_rows=2048)
    items = [(p["seq"], p["coords"], p["q_aln"], p["t_aln"]) for p in prots]

    class Aln:
        def __init__(self, p):
            self.coords, self.gapped_sequence, self.gapped_target = p["coords"], p["q_aln"], p["t_aln"]

    stream = AlignmentStream(eng, batch_size=10, max_rows=2048)
    firsts, got = [], {"mf": [], "cc": []}
    for first, res in stream.run(items[:25] + [Aln(p) for p in prots[25:]]):
        firsts.append(first)
        for m in got:
            got[m].append(res[m])
    assert firsts == [0, 10, 20, 30, 40, 50]
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots], max_rows=2048)
    ref = eng.run_alignments(pk)
    for m in got:
        assert np.array_equal(np.concatenate(got[m]), ref[m])
    arrival = AlignmentStream(eng, batch_size=10, max_rows=2048, sort_by_length=False).run_all(items)      # batches packed in arrival order:
    assert all(np.array_equal(arrival[m], ref[m]) for m in got)                                              # (the default sorts each by length) the same bits
    bad = list(items)
    bad[37] = ("ACDJ", bad[37][1][:4], "ACDJ", "ACDJ")
    with pytest.raises(ValueError, match="Invalid character in sequence: J"):
        stream.run_all(bad)


def test_alignment_stream_recovers_from_csr_overflow(mf):
    """A batch denser than the planned CSR capacity is redone with a larger one; the stream still delivers every batch in order."""
    from mDeepFRI.batch import PackedProteins
    from mDeepFRI.stream import AlignmentStream
    w, pred = mf
    prots = synthetic.synthetic_proteins(seed=41, count=12, length=(60, 120))
    items = [(p["seq"], p["coords"], p["q_aln"], p["t_aln"]) for p in prots]
    eng = _engine({"mf": pred}, max_rows=1024, nnz_per_row=2)            # far too small: ~12 neighbours per residue
    got = AlignmentStream(eng, batch_size=5, max_rows=1024).run_all(items)["mf"]
    ref = _engine({"mf": pred}, max_rows=1024).run_alignments(PackedProteins.pack(*zip(*items), max_rows=1024))["mf"]
    assert np.array_equal(got, ref)


def test_forward_pass_from_several_threads(mf, cc):
    """The reference's session.run is thread-safe and ctypes releases the GIL during the call: concurrent forward_pass calls on
    ONE predictor (shared device scratch, NULL stream) and on two predictors must return exactly the single-threaded results."""
    import threading
    (wm, pm), (wc, pc) = mf, cc
    rng = np.random.default_rng(17)
    cases = []
    for L in (40, 300, 97, 512, 33, 200, 150, 64):
        cases.append((synthetic.random_sequence(rng, L), orc.calculate_contact_map(synthetic.random_walk_coords(rng, L), 6.0)))
    expect = [(pm.forward_pass(s, c), pc.forward_pass(s, c)) for s, c in cases]
    errors = []

    def worker(tid):
        try:
            for rep in range(6):
                for k in range(len(cases)):
                    i = (k + tid * 3 + rep) % len(cases)
                    pred, ref = (pm, expect[i][0]) if (tid + k) % 3 else (pc, expect[i][1])
                    if not np.array_equal(pred.forward_pass(*cases[i]), ref):
                        errors.append((tid, rep, i))
        except Exception as e:      # noqa: BLE001
            errors.append((tid, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


def test_aggregation_kernel_is_chosen_per_protein(mf):
    """Which kernel aggregates a protein -- the matrix-pipe product on the contact BITS (binary map, at most 512 residues) or the CSR gather
    (longer proteins, maps with other values) -- depends on that protein alone, so the batched dense-map path, the per-call API and,
    for maps made from coordinates, the fused path agree bit for bit whatever else is in the batch.  Cases: a symmetric 6 A map, the
    reference's own test recipe (np.random.randint(0, 2, (L, L)): binary, NOT symmetric, weight_convert notebook cell 1), a float map
    with entries other than 0 / 1 (gather), a binary map with an odd DIAGONAL (the kernels force the diagonal to 1: still binary), a
    600-residue protein (gather), all in one batch; each vs the oracle and batch == per call."""
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    w, pred = mf
    rng = np.random.default_rng(31)
    seqs, maps = [], []
    for kind, L in (("sym", 100), ("randint", 300), ("float", 96), ("diag", 77), ("long", 600), ("sym", 512), ("randint", 17)):
        seqs.append(synthetic.random_sequence(rng, L))
        if kind == "randint":
            cm = rng.integers(0, 2, size=(L, L)).astype(np.int32)
        else:
            cm = orc.calculate_contact_map(synthetic.random_walk_coords(rng, L), 6.0).astype(np.int32)
        if kind == "float":
            cm = cm.astype(np.float32) * rng.choice(np.array([0.5, 1.0, 2.0], dtype=np.float32), size=(L, L))
        if kind == "diag":
            cm = cm.copy()
            cm[np.arange(L), np.arange(L)] = 7
        maps.append(cm)
    eng = HotPathEngine({"mf": pred}, device=0, max_rows=4096)
    out = eng.forward_dense(eng.upload(PackedProteins.pack(seqs, max_rows=4096)), [m.astype(np.float32) for m in maps])["mf"].cpu().numpy()
    out_i32 = eng.forward_dense(eng.upload(PackedProteins.pack(seqs[:2] + seqs[3:], max_rows=4096)), maps[:2] + maps[3:])["mf"].cpu().numpy()
    for i, (s, cm) in enumerate(zip(seqs, maps)):
        y = pred.forward_pass(s, cm)
        assert np.max(np.abs(y - gcn_oracle.gcn_forward(w, s, cm))) < TOL, i
        assert np.array_equal(out[i], pred.forward_pass(s, cm.astype(np.float32))), i      # batch == per call, bitwise (float32 maps)
        if i != 2:
            assert np.array_equal(out_i32[i - (i > 2)], y), i                             # ... and for int32 maps
    assert np.array_equal(out[0], out_i32[0])                                              # the dtype of a binary map changes nothing


def _pipe_error_script() -> str:
    from conftest import ROOT
    return (
        "import sys, os, json; ROOT = %r\n"
        "for d in ('metagenomic-deepfri_amd', 'oracle', ''): sys.path.insert(0, os.path.join(ROOT, d))\n"
        "import numpy as np\n"
        "import cmap_oracle as orc, gcn_oracle\n"
        "from mdfri_testkit import synthetic\n"
        "from mDeepFRI import _hip\n"
        "from mDeepFRI.batch import HotPathEngine, PackedProteins\n"
        "from mDeepFRI.predict import Predictor\n"
        "w = synthetic.glorot_gcn_weights(seed=5, n_terms=96)\n"
        "pred = Predictor('syn', weights=w)\n"
        "prots = synthetic.synthetic_proteins(seed=71, count=36, length=(380, 520))\n"      # > 12 288 rows: the batch runs the 256 x 256 tile kernel
        "pk = PackedProteins.pack([q['seq'] for q in prots], [q['coords'] for q in prots], [q['q_aln'] for q in prots], [q['t_aln'] for q in prots], max_rows=32768)\n"
        "assert pk.chunks[0].rows > 12288\n"
        "out = HotPathEngine({'a': pred}, device=0, max_rows=32768).run_alignments(pk)['a']\n"
        "eb = ec = 0.0; same = True\n"
        "for i in (0, 7, 19, 35):\n"
        "    q = prots[i]\n"
        "    cm = orc.build_align_contact_map(q['coords'], q['q_aln'], q['t_aln'], 6.0, 2)\n"
        "    ref = gcn_oracle.gcn_forward(w, q['seq'], cm, dtype=np.float64)\n"
        "    y = pred.forward_pass(q['seq'], cm)\n"                                          # one protein: the wave-per-tile kernel
        "    same = same and bool(np.array_equal(y, out[i]))\n"
        "    eb = max(eb, float(np.max(np.abs(out[i] - ref)))); ec = max(ec, float(np.max(np.abs(y - ref))))\n"
        "print('RESULT', json.dumps({'pipe': _hip.lib().mdf_hw_pipe().decode(), 'batch_err': eb, 'percall_err': ec, 'bitwise': same}))\n" % ROOT)


def test_bf16x6_products_are_at_least_as_accurate_as_the_fp32_instruction():
    """The graph-convolution products H.W run as BF16x6 on the bf16 matrix pipe by default (csrc/gcn.hip: every fp32 operand = three bf16
    terms, six term products per fp32 product, fp32 accumulation); MDFRI_HW_PIPE=f32 keeps them on v_mfma_f32_32x32x2_f32.  Both, in a
    process of their own, against the FLOAT64 oracle on the same proteins: the default path's error is not above the fp32 instruction's
    (beyond 25 %% + 1e-7 of slack for different accumulation orders), both are two orders inside the 1e-4 score tolerance, and in both
    the batch (256 x 256 tile kernel) equals the per-call API (wave-per-tile kernel) bit for bit."""
    import json
    import subprocess
    import sys
    res = {}
    for pipe in ("bf16x6", "f32"):
        env = {k: v for k, v in os.environ.items() if k != "MDFRI_HW_PIPE"}
        if pipe == "f32":
            env["MDFRI_HW_PIPE"] = "f32"
        out = subprocess.run([sys.executable, "-c", _pipe_error_script()], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        res[pipe] = json.loads(out.stdout.split("RESULT", 1)[1])
        assert res[pipe]["pipe"] == pipe and res[pipe]["bitwise"] is True, res[pipe]
        assert res[pipe]["batch_err"] < 2e-6 and res[pipe]["percall_err"] < 2e-6, res[pipe]
    assert res["bf16x6"]["batch_err"] <= 1.25 * res["f32"]["batch_err"] + 1e-7, res


def test_f16x3_pipe_opt_in_is_fp32_class_and_bitwise_across_kernels():
    """MDFRI_HW_PIPE=f16x3 (opt-in, csrc/gcn.hip "F16x3"): the GraphConv layers' H.W products from THREE fp16 term products (operands scaled by a
    power of two, two fp16 terms each: 22 of 24 bits).  In a process of its own against the FLOAT64 oracle on the proteins of the pipe test
    above: the library reports the pipe, the batch (k_gemm_f16x3, 256 x 256 tiles) equals the per-call API (k_gemm_f16x3_small, a wave per
    tile) bit for bit, and the error stays two orders inside the 1e-4 score tolerance -- within 1.5 x + 1e-7 of the default pipe's on the
    same proteins (it is not held to "at most the fp32 instruction's": the two-term split is not exact)."""
    import json
    import subprocess
    import sys
    res = {}
    for pipe in ("bf16x6", "f16x3"):
        env = {k: v for k, v in os.environ.items() if k != "MDFRI_HW_PIPE"}
        if pipe != "bf16x6":
            env["MDFRI_HW_PIPE"] = pipe
        out = subprocess.run([sys.executable, "-c", _pipe_error_script()], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[pipe] = json.loads(out.stdout.split("RESULT", 1)[1])
        assert res[pipe]["pipe"] == pipe and res[pipe]["bitwise"] is True, res[pipe]
        assert res[pipe]["batch_err"] < 2e-6 and res[pipe]["percall_err"] < 2e-6, res[pipe]
    assert res["f16x3"]["batch_err"] <= 1.5 * res["bf16x6"]["batch_err"] + 1e-7, res
    assert res["f16x3"] != res["bf16x6"]      # the switch did change the arithmetic


def test_f16x3_pipe_turns_an_activation_beyond_its_range_into_nan_scores_not_into_wrong_ones():
    """The documented limit of the opt-in pipe (include/mdfri.h, mdf_hw_pipe): the activations' scale is the constant 2^3, so an aggregated
    activation of magnitude >= 8 190 becomes inf in the split and the protein's scores NaN -- loud.  A model whose first GraphConv weights are
    blown up by 1e6 produces such activations: under MDFRI_HW_PIPE=f16x3 every score of the protein is NaN, on the default pipe the same call
    returns finite scores (bf16 has fp32's exponent range).  Each pipe in a process of its own."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    script = (
        "import sys, os, json; ROOT = %r\n"
        "for d in ('metagenomic-deepfri_amd', 'oracle', ''): sys.path.insert(0, os.path.join(ROOT, d))\n"
        "import numpy as np\n"
        "import cmap_oracle as orc\n"
        "from mdfri_testkit import synthetic\n"
        "from mDeepFRI import _hip\n"
        "from mDeepFRI.predict import Predictor\n"
        "w = synthetic.glorot_gcn_weights(seed=3, n_terms=64)\n"
        "w['W_gc1'] = (w['W_gc1'] * np.float32(1e6)).astype(np.float32)\n"
        "p = synthetic.synthetic_proteins(seed=5, count=1, length=300)[0]\n"
        "cm = orc.build_align_contact_map(p['coords'], p['q_aln'], p['t_aln'], 6.0, 2)\n"
        "y = Predictor('big', weights=w).forward_pass(p['seq'], cm)\n"
        "print('RESULT', json.dumps({'pipe': _hip.lib().mdf_hw_pipe().decode(), 'nan': int(np.isnan(y).sum()), 'finite': int(np.isfinite(y).sum()), 'n': int(y.size)}))\n" % ROOT)
    res = {}
    for pipe in ("bf16x6", "f16x3"):
        env = {k: v for k, v in os.environ.items() if k != "MDFRI_HW_PIPE"}
        if pipe != "bf16x6":
            env["MDFRI_HW_PIPE"] = pipe
        out = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[pipe] = json.loads(out.stdout.split("RESULT", 1)[1])
        assert res[pipe]["pipe"] == pipe, res
    assert res["bf16x6"]["finite"] == res["bf16x6"]["n"], res
    assert res["f16x3"]["nan"] == res["f16x3"]["n"], res


def test_agg_prepare_writes_the_contact_bits_as_byte_tiles():
    """mdf_agg_prepare_dev (round 6): beside d_j and the populated-block words, the contact bits once more in the order the matrix-pipe
    aggregation loads them (mdfri.h mdf_agg_desc.tiles): 16-row group g x 256-column chunk c of a protein at ((g nch + c) 512), byte
    ((i mod 16) 2 + h) 16 + cb = the bits of row i, columns 256 c + 16 cb + 8 h .. + 7.  Random symmetric-free bit matrices for proteins of
    80 .. 1 024 residues (and one too short, one too long: no tiles, no block words), rebuilt in numpy from the same masks."""
    import ctypes
    import torch
    from mDeepFRI import _hip
    L = _hip.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(12)
    lens = [80, 97, 256, 257, 300, 512, 513, 1000, 1024, 64, 1500]
    Lq = np.array(lens, dtype=np.int32)
    row_off = np.zeros(len(lens) + 1, dtype=np.int32)
    R = int(L.mdf_layout_rows(_hip.ptr(Lq), len(lens), _hip.ptr(row_off)))
    max_len = int(Lq.max())
    W = (max_len + 63) // 64
    masks = np.zeros((R, W), dtype=np.uint64)
    counts = np.zeros(R, dtype=np.int32)
    dense = []
    for p, n in enumerate(lens):
        A = (rng.random((n, n)) < 0.04)
        A |= np.abs(np.subtract.outer(np.arange(n), np.arange(n))) <= 2
        dense.append(A)
        bits = np.zeros((n, W * 64), dtype=bool)
        bits[:, :n] = A
        words = np.packbits(bits.reshape(n, W, 64)[:, :, ::-1], axis=2).view(">u8").reshape(n, W).astype(np.uint64)   # bit j of word w = column 64 w + j
        masks[row_off[p]:row_off[p] + n] = words
        counts[row_off[p]:row_off[p] + n] = A.sum(axis=1)
    Wt = int(L.mdf_agg_tile_row_bytes(max_len))
    assert Wt == 128
    up = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    d_masks, d_counts, d_ro, d_lq = up(masks.view(np.int64)), up(counts), up(row_off), up(Lq)
    dinv = torch.full((R,), -1.0, dtype=torch.float32, device=dev)
    blk = torch.zeros((len(lens), 32), dtype=torch.int64, device=dev)
    tiles = torch.full((R * Wt,), 0xEE, dtype=torch.uint8, device=dev)
    p_ = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    _hip.check(L.mdf_agg_prepare_dev(p_(d_masks), W, p_(d_counts), p_(d_ro), p_(d_lq), len(lens), R, p_(dinv), p_(blk), p_(tiles), Wt, None))
    torch.cuda.synchronize()
    th, bh, dh = tiles.cpu().numpy(), blk.cpu().numpy().view(np.uint64), dinv.cpu().numpy()
    for p, n in enumerate(lens):
        r0, A = int(row_off[p]), dense[p]
        assert np.array_equal(dh[r0:r0 + n], (1.0 / (np.float32(1e-6) + np.sqrt(A.sum(axis=1).astype(np.float32)))).astype(np.float32)), p
        if n < 80 or n > 1024:
            assert np.all(th[r0 * Wt:(r0 + n) * Wt] == 0xEE) and not bh[p].any()       # outside the matrix-pipe lengths: untouched
            continue
        npad, nch = (n + 15) // 16 * 16, (n + 255) // 256
        Ap = np.zeros((npad, nch * 256), dtype=bool)
        Ap[:n, :n] = A
        region = th[r0 * Wt:r0 * Wt + (npad // 16) * nch * 512].reshape(npad // 16, nch, 16, 2, 16)     # [group][chunk][row in group][half][column block]
        want = np.packbits(Ap.reshape(npad // 16, 16, nch, 16, 2, 8)[..., ::-1], axis=5)[..., 0]           # [g][i][c][cb][h] -> byte, bit k = column ... + k
        assert np.array_equal(region, want.transpose(0, 2, 1, 4, 3)), (p, n)
        for b in range((n + 31) // 32):      # bit c of blk[p][b]: rows [32 b, 32 b + 32) have a contact in columns [16 c, 16 c + 16)
            rows = Ap[32 * b:32 * b + 32]
            exp = sum(1 << c for c in range(nch * 16) if rows[:, 16 * c:16 * c + 16].any())
            assert int(bh[p, b]) == exp, (p, b)


def test_dense_map_path_replans_a_batch_planned_with_the_fused_paths_chunks():
    """HotPathEngine.forward_dense stages a chunk's maps in pinned + device slots of chunk rows x L x 4 B each; a batch planned with the
    fused path's default chunk (262 144 rows) is re-planned with DENSE_CHUNK_ROWS (65 536) -- same scores, bit for bit, as the same maps through
    a batch planned small from the start and as the fused path."""
    import cmap_oracle
    from mDeepFRI import batch
    from mDeepFRI.batch import HotPathEngine, PackedProteins
    from mDeepFRI.predict import Predictor
    from mdfri_testkit import synthetic
    w = synthetic.glorot_gcn_weights(seed=3, n_terms=55)
    eng = HotPathEngine({"a": Predictor("syn", weights=w)}, device=0)
    prots = synthetic.synthetic_proteins(seed=77, count=300, length=(200, 330), indel_rate=0.03)      # ~80 000 padded rows: two chunks of 65 536, one of the default
    maps = [cmap_oracle.build_align_contact_map(p["coords"], p["q_aln"], p["t_aln"], 6.0, 2) for p in prots]
    seqs = [p["seq"] for p in prots]
    big = eng.upload(PackedProteins.pack(seqs))
    assert big.packed.max_chunk_rows > batch.DENSE_CHUNK_ROWS and len(big.packed.chunks) == 1
    got = eng.forward_dense(big, maps)["a"].cpu().numpy()
    assert len(big._dense.packed.chunks) == 2 and big._dense.packed.max_chunk_rows <= batch.DENSE_CHUNK_ROWS
    small = eng.forward_dense(eng.upload(PackedProteins.pack(seqs, max_rows=batch.DENSE_CHUNK_ROWS)), maps)["a"].cpu().numpy()
    fused = eng.run_alignments(PackedProteins.pack(seqs, [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots]))["a"]
    assert np.array_equal(got, small) and np.array_equal(got, fused)
