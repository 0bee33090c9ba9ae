"""GPU parity: contact-map kernels (through the C ABI / the drop-in Python API) vs the golden vectors of the
compiled reference and vs the C oracle.  Bit-exact everywhere (integer / index work and f32 bit patterns)."""
import hashlib
import os

import numpy as np
import pytest

import cmap_oracle as orc
from conftest import gstr
from mdfri_testkit import synthetic

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class Aln:
    def __init__(self, coords, q, t):
        self.coords, self.gapped_sequence, self.gapped_target = coords, q, t
        self.target_name, self.query_name = "target.pdb", "query"


def test_pairwise_goldens_bit_exact(cmap_golden):
    from mDeepFRI.contact_map_utils import pairwise_sqeuclidean
    keys = [k[:-2] for k in cmap_golden.files if k.startswith("pairwise/") and k.endswith("/X")]
    assert len(keys) >= 8
    for k in keys:
        X = np.ascontiguousarray(cmap_golden[k + "/X"])
        D = pairwise_sqeuclidean(X)
        exp = cmap_golden[k + "/D"]
        assert D.dtype == np.float32 and D.shape == exp.shape, k
        assert np.array_equal(D.view(np.uint32), exp.view(np.uint32)), k


def test_pairwise_reference_kat():
    # reference mDeepFRI/tests/test_contact_map_utils.py:15-25
    from mDeepFRI.contact_map_utils import pairwise_sqeuclidean
    np.random.seed(42)
    expected = np.array([[0, 1.01354558, 0.12442072], [1.01354558, 0, 0.99467713], [0.12442072, 0.99467713, 0]], dtype=np.float32)
    result = pairwise_sqeuclidean(np.random.rand(3, 3).astype(np.float32))
    assert np.allclose(result, expected)


def test_pairwise_rejects_wrong_buffers():
    from mDeepFRI.contact_map_utils import pairwise_sqeuclidean
    with pytest.raises(ValueError):
        pairwise_sqeuclidean(np.zeros((3, 3), dtype=np.float64))
    with pytest.raises(ValueError):
        pairwise_sqeuclidean(np.asfortranarray(np.zeros((3, 4), dtype=np.float32)))


@pytest.mark.parametrize("n", [127, 128, 129, 1024, 2048])
def test_pairwise_random_vs_oracle(n):
    from mDeepFRI.contact_map_utils import pairwise_sqeuclidean
    X = synthetic.random_walk_coords(np.random.default_rng(n), n)
    D = pairwise_sqeuclidean(X)
    assert np.array_equal(D.view(np.uint32), orc.pairwise_sqeuclidean(X).view(np.uint32))
    assert np.array_equal(D, D.T) and np.all(np.diag(D) == 0)


def test_pairwise_nan_rows_keep_zero_diagonal():
    from mDeepFRI.contact_map_utils import pairwise_sqeuclidean
    X = np.array([[0, 0, 0], [np.nan, 1, 2], [3, 4, np.inf]], dtype=np.float32)
    D, E = pairwise_sqeuclidean(X), orc.pairwise_sqeuclidean(X)
    # NaN lands in the same cells; its sign/payload bits are platform-specific (x86 SSE and GCN propagate NaN operands
    # differently) and carry no meaning: every later use is `D < thr`, false for any NaN.  All other cells bit-exact.
    assert np.array_equal(np.isnan(D), np.isnan(E))
    ok = ~np.isnan(E)
    assert np.array_equal(D.view(np.uint32)[ok], E.view(np.uint32)[ok])
    assert np.all(np.diag(D) == 0)


def test_special_values_bit_exact_vs_the_compiled_reference(cmap_special_golden):
    """subnormal d^2 (points 1e-20 apart), d^2 that overflows to inf, -0.0 / 1e18-scale / NaN / inf coordinates and pairs exactly
    on the threshold, against fixtures made by the compiled reference (tests/golden/make_special_golden.py): the per-call distance kernel bit
    for bit (NaN cells: NaN-ness), the thresholded map at six thresholds (one float32 step either side of 6 A, 1e-19 A whose square is
    subnormal, 3e19 A whose square overflows the float32 the comparison is made in, and 0), the per-call aligned map, and the batched contact
    stage (k_cmap_bits: packed fp32 on row pairs away from the diagonal, per-row code next to it) through build_align_contact_maps.  The
    kernels run with float32 denormals on (hipcc's default for gfx950; no flush-to-zero flag in csrc/Makefile): no divergence to document."""
    from conftest import same_float_bits, special_cases
    from mDeepFRI.alignment import AlignmentResult
    from mDeepFRI.batch import build_align_contact_maps
    from mDeepFRI.bio_utils import build_align_contact_map, calculate_contact_map
    from mDeepFRI.contact_map_utils import pairwise_sqeuclidean
    alns = []
    for name, X, D_bits, maps, aligned in special_cases(cmap_special_golden):
        assert same_float_bits(pairwise_sqeuclidean(X), D_bits), name
        for thr, cm in maps:
            got = calculate_contact_map(X, thr)
            assert got.dtype == np.int32 and np.array_equal(got, cm), (name, thr, int(got.sum()), int(cm.sum()))
        seq = "A" * X.shape[0]
        for gen, want in aligned.items():
            _, got = build_align_contact_map(Aln(X, seq, seq), 6.0, gen)
            assert np.array_equal(got, want), (name, gen)
        a = AlignmentResult(query_name=name, query_sequence=seq, target_name="t", target_sequence=seq, alignment="")
        a.gapped_sequence, a.gapped_target, a.coords = seq, seq, X
        alns.append((a, aligned))
    for gen in (0, 2):      # the batched stage, the two sets side by side in one launch (and once more behind a long protein: another chunk geometry)
        for extra in ([], [800]):
            batch = [a for a, _ in alns]
            for L in extra:
                rng = np.random.default_rng(L)
                s2 = synthetic.random_sequence(rng, L)
                b = AlignmentResult(query_name="pad", query_sequence=s2, target_name="t", target_sequence=s2, alignment="")
                b.gapped_sequence, b.gapped_target, b.coords = s2, s2, synthetic.random_walk_coords(rng, L)
                batch = [b] + batch
            res = build_align_contact_maps(batch, threshold=6.0, generated_contacts=gen, device=0)
            for (a, aligned), (_, got) in zip(alns, res[len(extra):]):
                assert got.dtype == np.int32 and np.array_equal(got, aligned[gen]), (a.query_name, gen, extra)
        # ... and the fused path's contact stage (k_cmap_bits + k_cmap_fill_rows): the CSR holds exactly the reference's contacts
        _assert_csr_holds_exactly_these_maps([(a.query_name, a.coords, a.gapped_sequence, a.gapped_target, aligned[gen]) for a, aligned in alns], 6.0, gen)


def test_align_golden_cases_bit_exact(cmap_golden):
    from mDeepFRI.contact_map_utils import align_contact_map
    for n in [str(x) for x in cmap_golden["index/align"]]:
        out = align_contact_map(gstr(cmap_golden[n + "/q"]), gstr(cmap_golden[n + "/t"]), cmap_golden[n + "/pairs"],
                                int(cmap_golden[n + "/gen"]))
        exp = cmap_golden[n + "/out"]
        assert out.dtype == np.int32 and out.shape == exp.shape, n
        assert np.array_equal(out, exp), n


def test_align_rejects_int64_pairs():
    from mDeepFRI.contact_map_utils import align_contact_map
    with pytest.raises(ValueError):
        align_contact_map("AB", "AB", np.array([[0, 1]], dtype=np.int64))


def test_chain_golden_cases(cmap_golden):
    """coords -> sparse -> aligned, per-call API, against the reference's outputs (sha256 of the exact bytes)."""
    from mDeepFRI.bio_utils import build_align_contact_map, calculate_contact_map
    from mDeepFRI.contact_map_utils import align_contact_map, pairwise_sqeuclidean
    for n in [str(x) for x in cmap_golden["index/chain"]]:
        coords = np.ascontiguousarray(cmap_golden[n + "/coords"])
        q, t, gen = gstr(cmap_golden[n + "/q"]), gstr(cmap_golden[n + "/t"]), int(cmap_golden[n + "/gen"])
        assert sha(pairwise_sqeuclidean(coords)) == gstr(cmap_golden[n + "/sha_D"]), n
        sparse = calculate_contact_map(coords, 6.0, mode="sparse")
        assert sparse.dtype == np.int32 and sha(sparse) == gstr(cmap_golden[n + "/sha_sparse"]), n
        out = align_contact_map(q, t, sparse, gen)
        assert sha(out) == gstr(cmap_golden[n + "/sha_out"]), n
        _, fused = build_align_contact_map(Aln(coords, q, t), 6.0, gen)
        assert sha(fused) == gstr(cmap_golden[n + "/sha_out"]), n


def test_calculate_contact_map_matrix_mode_and_threshold_edge():
    from mDeepFRI.bio_utils import calculate_contact_map
    coords = np.array([[0, 0, 0], [6, 0, 0], [0, 5.9999, 0], [3, 4, 0]], dtype=np.float32)
    cm = calculate_contact_map(coords, 6.0)
    assert np.array_equal(cm, orc.calculate_contact_map(coords, 6.0))
    assert cm[0, 1] == 0 and cm[0, 2] == 1 and cm[0, 3] == 1  # strict '<': exactly 6 A is not a contact
    for thr in (0.0, 3.8, 4.5, 6.1, 10.0):
        X = synthetic.random_walk_coords(np.random.default_rng(5), 200)
        assert np.array_equal(calculate_contact_map(X, thr), orc.calculate_contact_map(X, thr)), thr
        assert np.array_equal(calculate_contact_map(X, thr, mode="sparse"), orc.calculate_contact_map(X, thr, mode="sparse")), thr


def test_build_align_missing_coords_returns_none():
    from mDeepFRI.bio_utils import build_align_contact_map
    a = Aln(None, "AC", "AC")
    assert build_align_contact_map(a) == (a, None)


def test_fuzz_align_and_fused_vs_oracle():
    from mDeepFRI.bio_utils import build_align_contact_map
    from mDeepFRI.contact_map_utils import align_contact_map
    rng = np.random.default_rng(2024)
    for it in range(40):
        L = int(rng.integers(1, 300))
        seq = synthetic.random_sequence(rng, L)
        q, t, lt = synthetic.mutate_alignment(rng, seq, float(rng.choice([0.0, 0.05, 0.3])))
        coords = synthetic.random_walk_coords(rng, lt).reshape(-1, 3)
        gen = int(rng.integers(0, 5))
        thr = float(rng.choice([4.0, 6.0, 8.0]))
        _, fused = build_align_contact_map(Aln(coords, q, t), thr, gen)
        assert np.array_equal(fused, orc.build_align_contact_map(coords, q, t, thr, gen)), it
        pairs = rng.integers(-2, lt + 3, size=(int(rng.integers(0, 5 * L)), 2)).astype(np.int32)
        assert np.array_equal(align_contact_map(q, t, pairs, gen), orc.align_contact_map(q, t, pairs, gen)), it


def test_fused_handles_coords_shorter_and_longer_than_alignment():
    from mDeepFRI.bio_utils import build_align_contact_map
    rng = np.random.default_rng(8)
    seq = synthetic.random_sequence(rng, 50)
    q, t, lt = synthetic.mutate_alignment(rng, seq, 0.1)
    for n_coords in (lt - 7, lt + 9):
        coords = synthetic.random_walk_coords(rng, n_coords)
        _, fused = build_align_contact_map(Aln(coords, q, t), 6.0, 2)
        assert np.array_equal(fused, orc.build_align_contact_map(coords, q, t, 6.0, 2)), n_coords


@pytest.mark.parametrize("L", [1024, 2048])
def test_full_size_properties(L):
    """Size-independent properties at BASELINE.json sizes: symmetry for coords-derived maps, unit diagonal,
    identity alignment == thresholded distance map, idempotence through sparsify -> align."""
    from mDeepFRI.bio_utils import build_align_contact_map, calculate_contact_map
    from mDeepFRI.contact_map_utils import align_contact_map
    rng = np.random.default_rng(L)
    seq = synthetic.random_sequence(rng, L)
    coords = synthetic.random_walk_coords(rng, L)
    _, cm = build_align_contact_map(Aln(coords, seq, seq), 6.0, 2)
    assert cm.shape == (L, L) and np.array_equal(cm, cm.T) and np.all(np.diag(cm) == 1)
    assert np.array_equal(cm, calculate_contact_map(coords, 6.0))
    sparse = calculate_contact_map(coords, 6.0, mode="sparse")
    assert np.all(np.diff(sparse[:, 0].astype(np.int64) * L + sparse[:, 1]) > 0)  # row-major sorted, unique
    assert np.array_equal(align_contact_map(seq, seq, sparse), cm)
    assert np.array_equal(cm, orc.build_align_contact_map(coords, seq, seq, 6.0, 2))


def test_contact_map_classes_reference_tests():
    # reference mDeepFRI/tests/test_conctact_map.py:8-41
    from mDeepFRI.contact_map import CAlphaCoordinates
    coords = np.array([[1, 2, 3], [4, 5, 6]])
    assert np.array_equal(CAlphaCoordinates("test", coords).coords, coords)
    with pytest.raises(ValueError, match="Coordinates are not 3D."):
        CAlphaCoordinates("test", np.array([[1, 2], [3, 4]]))
    ac = CAlphaCoordinates("test", np.array([[0, 0, 0], [1, 1, 1]]))
    dm = ac.calculate_distance_map(distance="sqeuclidean")
    assert np.allclose(np.sqrt(dm.distance_map), np.array([[0, np.sqrt(3)], [np.sqrt(3), 0]], dtype=np.float32))
    with pytest.raises(NotImplementedError, match="Distance metric not implemented."):
        ac.calculate_distance_map(distance="euclidean")
    cm = CAlphaCoordinates("test", np.array([[0, 0, 0], [5, 0, 0], [10, 0, 0]])).calculate_contact_map(threshold=6.0)
    assert np.array_equal(cm.cmap, np.array([[1, 1, 0], [1, 1, 1], [0, 1, 1]]))
    assert np.array_equal(cm.sparsify(), np.argwhere(cm.cmap == 1).astype(np.int32))


@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.float16, np.int32, np.int64, np.uint8])
def test_distance_map_calculate_contacts_accepts_any_real_dtype(dtype):
    """reference contact_map.py:74 is `(distance_map < threshold).astype(int32)` for a map of any dtype; expected values are
    NumPy's own comparison (float32 maps compare in float32, everything else in float64)."""
    from mDeepFRI.contact_map import DistanceMap
    rng = np.random.default_rng(12)
    a = rng.random((70, 70)) * 80
    d = ((a + a.T) / 2)
    np.fill_diagonal(d, 0)
    d = d.astype(dtype)
    d[3, 5] = d[5, 3] = 36        # exactly on the threshold: strict '<' keeps it out
    for thr in (36.0, 36.5, 6.0**2 + 1e-9):
        got = DistanceMap(d).calculate_contacts(thr).cmap
        assert got.dtype == np.int32 and np.array_equal(got, (d < thr).astype(np.int32)), (dtype, thr)
    # a float64 map whose entries differ from the threshold only beyond float32 precision: must compare in float64
    e = np.array([[0.0, 36.0 - 1e-12], [36.0 - 1e-12, 0.0]])
    assert np.array_equal(DistanceMap(e).calculate_contacts(36.0).cmap, [[1, 1], [1, 1]])
    assert np.array_equal(DistanceMap(e.astype(np.float32)).calculate_contacts(36.0).cmap, [[1, 0], [0, 1]])


def test_seq2onehot_reference_tests_and_alphabet():
    # reference mDeepFRI/tests/test_predict.py:8-33
    from mDeepFRI.predict import seq2onehot
    r = seq2onehot("")
    assert r.shape == (0, 26) and r.dtype == np.float32
    assert np.all(seq2onehot("D") == np.array([[0, 1] + [0] * 24]))
    exp = np.zeros((4, 26), np.float32)
    exp[np.arange(4), np.arange(4)] = 1
    assert np.all(seq2onehot("-DGU") == exp)
    with pytest.raises(ValueError):
        seq2onehot("J")
    alpha = "-DGULNTKHYWCPVSOIEFXQABZRM"
    assert np.array_equal(seq2onehot(alpha), np.eye(26, dtype=np.float32))
    with pytest.raises(ValueError, match="Invalid character in sequence: j"):
        seq2onehot("ACjDE*")
    long = synthetic.random_sequence(np.random.default_rng(0), 3000)
    assert np.array_equal(seq2onehot(long), orc.seq2onehot(long))


def test_batched_build_align_contact_maps_matches_per_call():
    from mDeepFRI.batch import build_align_contact_maps
    prots = synthetic.synthetic_proteins(seed=3, count=37, length=(20, 400), indel_rate=0.07)
    alns = [Aln(p["coords"], p["q_aln"], p["t_aln"]) for p in prots]
    alns.insert(5, Aln(None, "AC", "AC"))
    res = build_align_contact_maps(alns, 6.0, 2, max_rows=4096)
    assert len(res) == len(alns) and res[5] == (alns[5], None)
    for a, (ra, cm) in zip(alns, res):
        assert ra is a
        if a.coords is not None:
            assert cm.dtype == np.int32
            assert np.array_equal(cm, orc.build_align_contact_map(a.coords, a.gapped_sequence, a.gapped_target, 6.0, 2))


def _assert_csr_holds_exactly_these_maps(cases, thr, gen):
    """The CSR the fused GCN path consumes (mdf_cmap_csr_dev: k_align_scan + k_cmap_bits + k_cmap_fill_rows) for `cases` = [(name, coords, q_aln,
    t_aln, expected aligned map)]: every row's column set == the expected map's row (plus the self loop GraphConv adds)."""
    import ctypes
    import torch
    from mDeepFRI import _hip
    from mDeepFRI.batch import DeviceBatch, PackedProteins, _p
    L = _hip.lib()
    dev = torch.device("cuda:0")
    pk = PackedProteins.pack([q.replace("-", "") for _, _, q, _, _ in cases], [c for _, c, _, _, _ in cases], [q for _, _, q, _, _ in cases],
                             [t for _, _, _, t, _ in cases], max_rows=65536, keep_order=True)   # (the stage is driven by hand below: plan position = batch position)
    assert len(pk.chunks) == 1
    db, R = DeviceBatch(pk, dev), pk.chunks[0].rows
    max_len, cap = int(pk.Lq.max()), R * 256
    ws = torch.empty(L.mdf_cmap_workspace_bytes(pk.B, R, max_len), dtype=torch.uint8, device=dev)
    rowptr = torch.empty(R + 1, dtype=torch.int32, device=dev)
    colidx = torch.empty(cap, dtype=torch.int32, device=dev)
    val = torch.empty(cap, dtype=torch.float32, device=dev)
    status = torch.zeros(4, dtype=torch.int32, device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _hip.check(L.mdf_cmap_csr_dev(_p(db.coords), _p(db.coord_off), _p(db.q_aln), _p(db.t_aln), _p(db.aln_off), _p(db.Lq), _p(db.chunk_row_off), pk.B, R,
                                  max_len, thr, gen, _p(rowptr), _p(colidx), _p(val), cap, _p(status), None, None, _p(ws), ws.numel(), st))
    torch.cuda.synchronize()
    assert status.tolist() == [0, 0, 0, 0]
    rp, ci = rowptr.cpu().numpy(), colidx.cpu().numpy()
    ro = pk.chunk_row_off
    for k, (n, _, _, _, out) in enumerate(cases):
        r0 = int(ro[k])
        for i in range(out.shape[0]):
            cols = ci[rp[r0 + i]:rp[r0 + i + 1]] - r0
            exp = np.flatnonzero(out[i] | (np.arange(out.shape[0]) == i))          # GraphConv adds the self loop
            assert np.array_equal(cols, exp), (n, i)


def test_other_thresholds_against_the_compiled_reference_goldens():
    """(10 A, 2) -- the operating point of the released `..._ca_10.0_...` models --, (4, 0), (8, 5), (10, 0), (7.5, 1): the per-call
    drop-in functions, the fused per-call build and the batched dense build, bit for bit against outputs of the COMPILED REFERENCE
    (tests/golden/cmap_thr_golden.npz, made by tests/golden/make_thr_golden.py); and the CSR the fused GCN path consumes holds
    exactly the reference's contacts (column indices per row)."""
    import ctypes
    import hashlib
    import torch
    from conftest import GOLDEN
    from mDeepFRI import _hip
    from mDeepFRI.batch import DeviceBatch, PackedProteins, _p, build_align_contact_maps
    from mDeepFRI.bio_utils import build_align_contact_map, calculate_contact_map
    from mDeepFRI.contact_map_utils import align_contact_map
    z = np.load(os.path.join(GOLDEN, "cmap_thr_golden.npz"))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()  # noqa: E731
    by_setting = {}
    for n in (str(x) for x in z["index"]):
        coords, q, t = z[n + "/coords"], gstr(z[n + "/q"]), gstr(z[n + "/t"])
        thr, gen = float(z[n + "/thr"]), int(z[n + "/gen"])
        sparse = calculate_contact_map(coords, thr, mode="sparse")
        assert sparse.dtype == np.int32 and sha(sparse) == gstr(z[n + "/sha_sparse"]), n
        out = align_contact_map(q, t, sparse, gen)
        assert sha(out) == gstr(z[n + "/sha_out"]), n
        assert np.array_equal(build_align_contact_map(Aln(coords, q, t), thr, gen)[1], out), n
        by_setting.setdefault((thr, gen), []).append((n, coords, q, t, out))
    L = _hip.lib()
    dev = torch.device("cuda:0")
    for (thr, gen), cases in by_setting.items():
        res = build_align_contact_maps([Aln(c, q, t) for _, c, q, t, _ in cases], thr, gen, max_rows=1024)
        for (n, _, _, _, out), (_, cm) in zip(cases, res):
            assert np.array_equal(cm, out), n
        _assert_csr_holds_exactly_these_maps(cases, thr, gen)


def test_csr_stage_flags_a_query_longer_than_max_len():
    """mdf_cmap_csr_dev sizes its contact-bit rows from `max_len`; a longer query must neither write out of its rows nor pass
    silently: status[2] carries its length."""
    import ctypes
    import torch
    from mDeepFRI import _hip
    from mDeepFRI.batch import DeviceBatch, PackedProteins, _p
    L = _hip.lib()
    prots = synthetic.synthetic_proteins(seed=3, count=2, length=200)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots])
    dev = torch.device("cuda:0")
    db = DeviceBatch(pk, dev)
    ch = pk.chunks[0]
    R = ch.rows
    for max_len, expect in ((200, 0), (100, 200)):
        ws = torch.empty(L.mdf_cmap_workspace_bytes(2, R, max_len), dtype=torch.uint8, device=dev)
        rowptr = torch.empty(R + 1, dtype=torch.int32, device=dev)
        colidx = torch.empty(R * 40, dtype=torch.int32, device=dev)
        val = torch.empty(R * 40, dtype=torch.float32, device=dev)
        status = torch.zeros(4, dtype=torch.int32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _hip.check(L.mdf_cmap_csr_dev(_p(db.coords), _p(db.coord_off), _p(db.q_aln), _p(db.t_aln), _p(db.aln_off), _p(db.Lq), _p(db.chunk_row_off), 2, R,
                                      max_len, 6.0, 2, _p(rowptr), _p(colidx), _p(val), R * 40, _p(status), None, None, _p(ws), ws.numel(), st))
        torch.cuda.synchronize()
        assert int(status[2].item()) == expect and int(status[0].item()) == 0


def test_contact_fill_forms_agree_bit_for_bit_on_the_raw_outputs():
    """The CSR / letter-sum step of the batched contact stage has two forms: k_cmap_fill_rows (eight lanes per row, the block's columns staged in
    LDS, the group scan inside) and, where that staging would not fit 64 KiB -- a batch whose longest query is beyond ~3 000 residues --,
    k_scan_groups + k_cmap_fill (a word per lane) for every protein of the launch.  The same proteins alone (first form) and in front of a
    3 300-residue query (second form, wider contact-bit rows): the raw outputs of mdf_cmap_csr_dev for THEIR rows -- row pointers, column
    indices, values, the layer-1 letter sums -- are the same bytes, over indels, proteins sharing 32-row blocks, several thresholds and
    generated-contact widths; and a capacity that overflows is flagged with clamped row pointers.  (Until round 6 an environment knob forced the
    second form, and a third kernel -- k_cmap_rows<COUNT>, rounds 1-4 -- stood behind another: experiments/r06_pruned_variants.patch.)"""
    import ctypes
    import torch
    from mDeepFRI import _hip
    from mDeepFRI.batch import DeviceBatch, PackedProteins, _p
    L = _hip.lib()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    long_one = synthetic.synthetic_proteins(seed=999, count=1, length=3300, indel_rate=0.02)[0]

    def stage(prots, thr, gen, seq_idx_all, cap_per_row=64, cap=None):
        pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots], [p["t_aln"] for p in prots],
                                 max_rows=1 << 17, keep_order=True)
        assert len(pk.chunks) == 1
        db, R = DeviceBatch(pk, dev), pk.chunks[0].rows
        max_len = int(pk.Lq.max())
        cap = cap or R * cap_per_row
        seq_idx = torch.from_numpy(seq_idx_all[:R].copy()).to(dev)
        ws = torch.zeros(L.mdf_cmap_workspace_bytes(pk.B, R, max_len), dtype=torch.uint8, device=dev)
        rowptr = torch.zeros(R + 1, dtype=torch.int32, device=dev)
        colidx, val = torch.zeros(cap, dtype=torch.int32, device=dev), torch.zeros(cap, dtype=torch.float32, device=dev)
        lsum, status = torch.zeros(R * 32, dtype=torch.float32, device=dev), torch.zeros(4, dtype=torch.int32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _hip.check(L.mdf_cmap_csr_dev(_p(db.coords), _p(db.coord_off), _p(db.q_aln), _p(db.t_aln), _p(db.aln_off), _p(db.Lq), _p(db.chunk_row_off), pk.B, R,
                                      max_len, thr, gen, _p(rowptr), _p(colidx), _p(val), cap, _p(status), _p(seq_idx), _p(lsum), _p(ws), ws.numel(), st))
        torch.cuda.synchronize()
        return R, rowptr.cpu().numpy(), colidx.cpu().numpy(), val.cpu().numpy(), lsum.cpu().numpy(), status.cpu().numpy(), pk

    for it, (thr, gen, lens) in enumerate(((6.0, 2, [40, 7, 300, 16, 17, 1, 513, 64, 31]), (8.0, 0, [200, 200, 90]), (4.5, 5, [33, 700, 15, 15, 260]),
                                           (6.0, 2, [int(x) for x in rng.integers(1, 400, size=40)]))):
        prots = [synthetic.synthetic_proteins(seed=50 * it + k, count=1, length=n, indel_rate=0.08 if n > 8 else 0.0)[0] for k, n in enumerate(lens)]
        letters = rng.integers(0, 26, size=1 << 17).astype(np.uint8)
        R, rp, ci, va, ls, status, pk = stage(prots, thr, gen, letters)
        assert status.tolist() == [0, 0, 0, 0]
        R2, rp2, ci2, va2, ls2, status2, pk2 = stage(prots + [long_one], thr, gen, letters)
        assert status2.tolist() == [0, 0, 0, 0] and R2 > R
        rows = int(pk2.chunk_row_off[len(prots)])         # the rows of the shared proteins: the long one starts at the next 16-row group
        assert rows <= R and np.array_equal(pk.chunk_row_off[:len(prots)], pk2.chunk_row_off[:len(prots)])
        n = int(rp[rows])
        assert n > 0 and np.array_equal(rp[:rows + 1], rp2[:rows + 1])
        assert np.array_equal(ci[:n], ci2[:n]) and np.array_equal(va[:n].view(np.uint32), va2[:n].view(np.uint32))
        assert np.array_equal(ls[:rows * 32].view(np.uint32), ls2[:rows * 32].view(np.uint32))
        assert np.count_nonzero(ls[:rows * 32]) > rows          # the letter sums are there
        # a capacity that overflows: flagged, nothing written past the arrays (the row pointers are clamped to the capacity)
        _, rp3, _, _, _, status3, _ = stage(prots, thr, gen, letters, cap=100)
        assert status3[0] == 1 and status3[1] >= n, status3.tolist()
