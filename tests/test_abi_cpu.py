"""CPU-only checks: the C-ABI library loads and exports every symbol include/mdfri.h declares, host-side helpers
behave, compute entry points fail loudly without a GPU, and the product never reaches into oracle/."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from mDeepFRI import _hip, weights
from mdfri_testkit import synthetic

HEADER = os.path.join(ROOT, "include", "mdfri.h")
PKG = os.path.join(ROOT, "metagenomic-deepfri_amd")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mdf_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = _hip.lib()
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in mdfri.h but not exported"
    assert sorted(_hip.SIGNATURES) == syms, "ctypes signature table out of sync with mdfri.h"
    assert b"gfx950" in lib.mdf_version()


def test_header_cites_reference_for_every_replaced_entry_point():
    src = open(HEADER).read()
    for ref in ("contact_map_utils.pyx:17-37", "contact_map_utils.pyx:44-117", "bio_utils.py:196-227", "bio_utils.py:348-385",
                "predict.pyx:17-48", "predict.pyx:75-102", "contact_map.py:64-75", "contact_map.py:88-95"):
        assert ref in src, ref


def test_compute_fails_loudly_without_gpu():
    if _hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    from mDeepFRI.contact_map_utils import align_contact_map, pairwise_sqeuclidean
    from mDeepFRI.predict import Predictor, seq2onehot
    with pytest.raises(_hip.MdfriError, match="no HIP device"):
        pairwise_sqeuclidean(np.zeros((4, 3), dtype=np.float32))
    with pytest.raises(_hip.MdfriError, match="no CPU fallback"):
        align_contact_map("AB", "AB", np.array([[0, 1]], dtype=np.int32))
    with pytest.raises(_hip.MdfriError):
        seq2onehot("ACD")
    with pytest.raises(_hip.MdfriError):
        Predictor("x.onnx", weights=synthetic.glorot_gcn_weights(0, 8, embed=64, gc_dims=(256, 256, 256), fc_dim=256))


def test_argument_validation_happens_before_the_device_is_touched():
    from mDeepFRI.contact_map_utils import align_contact_map, pairwise_sqeuclidean
    from mDeepFRI.predict import seq2onehot
    with pytest.raises(ValueError, match="dtype mismatch"):
        pairwise_sqeuclidean(np.zeros((3, 3), dtype=np.float64))      # reference: Cython buffer ValueError
    with pytest.raises(ValueError):
        pairwise_sqeuclidean(np.zeros((3,), dtype=np.float32))
    with pytest.raises(ValueError, match="dtype mismatch"):
        align_contact_map("AB", "AB", np.array([[0, 1]], dtype=np.int64))
    with pytest.raises(TypeError):
        align_contact_map(b"AB", "AB", np.zeros((0, 2), dtype=np.int32))
    with pytest.raises(TypeError):
        seq2onehot(b"ACD")
    assert seq2onehot("").shape == (0, 26)                            # reference test_predict.py:9-14, no GPU needed


def test_align_len_and_layout_rows_host_helpers():
    lib = _hip.lib()
    lq = ctypes.c_int64(-1)
    assert lib.mdf_align_len(b"A-C--D", b"ABCDEF", 6, lq) == 0 and lq.value == 3
    assert lib.mdf_align_len(b"", b"", 0, lq) == 0 and lq.value == 0
    L = np.array([1, 32, 33, 512, 100], dtype=np.int32)
    ro = np.zeros(6, dtype=np.int32)
    R = lib.mdf_layout_rows(_hip.ptr(L), 5, _hip.ptr(ro))
    # every protein starts on a MDF_GROUP_ROWS = 16 boundary, the total is a multiple of 128
    assert lib.mdf_group_rows() == 16
    assert list(ro[:5]) == [0, 16, 48, 96, 608] and R == ro[5] == 768 and R % 128 == 0
    assert lib.mdf_layout_rows(_hip.ptr(np.array([-1], dtype=np.int32)), 1, _hip.ptr(ro)) < 0
    assert "negative" in _hip.last_error()


def test_null_and_bad_arguments_are_rejected():
    lib = _hip.lib()
    assert lib.mdf_pairwise_sqeuclidean_f32(None, 4, 3, None, 1) == _hip.MDF_EINVAL
    assert lib.mdf_model_num_terms(None) == _hip.MDF_EINVAL
    h = ctypes.c_void_p()
    assert lib.mdf_model_load(b"/nonexistent/model.mdfw", 0, ctypes.byref(h)) == _hip.MDF_EIO
    assert lib.mdf_timing_read(b"nope", None, None) == _hip.MDF_EINVAL
    assert lib.mdf_timing_enable(0) == 0 and lib.mdf_timing_reset() == 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libmdfri_hip.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _hip.lib()


def test_weight_container_round_trip(tmp_path):
    w = synthetic.glorot_gcn_weights(seed=3, n_terms=17, embed=64, gc_dims=(256, 256, 256), fc_dim=128)
    p = tmp_path / "m.mdfw"
    weights.save_mdfw(str(p), w)
    r = weights.load_mdfw(str(p))
    assert sorted(r) == sorted(w) and all(np.array_equal(r[k], w[k]) for k in w)
    assert weights.validate(r) == {"embed": 64, "gc_dims": [256, 256, 256], "fc_dim": 128, "n_terms": 17, "lm_dim": 0, "embed_linear": False}
    assert weights.resolve_model_path(str(tmp_path / "m.onnx")) == str(p)
    with pytest.raises(FileNotFoundError, match="no such model file"):
        weights.resolve_model_path(str(tmp_path / "other.onnx"))
    bad = dict(w)
    bad["W_fc"] = bad["W_fc"][:-1]
    with pytest.raises(ValueError):
        weights.validate(bad)


def test_packing_and_chunk_plan():
    from mDeepFRI.batch import PackedProteins
    prots = synthetic.synthetic_proteins(seed=1, count=20, length=(10, 300), indel_rate=0.1)
    pk = PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                             [p["t_aln"] for p in prots], max_rows=1024)
    assert pk.B == 20 and pk.seq_off[-1] == sum(len(p["seq"]) for p in prots)
    assert pk.coord_off[-1] == sum(p["coords"].shape[0] for p in prots)
    covered = []
    plan_lq = pk.Lq if pk.order is None else pk.Lq[pk.order]       # the plan visits the proteins shortest first: its tables speak of plan positions
    assert np.all(np.diff(plan_lq) >= 0)
    for ch in pk.chunks:
        ro = pk.chunk_row_off[ch.row_off_pos:ch.row_off_pos + (ch.p1 - ch.p0) + 1]
        assert ro[0] == 0 and ro[-1] == ch.rows and ch.rows % 128 == 0 and np.all(ro[:-1] % 16 == 0)
        assert np.all(np.diff(ro)[:-1] >= plan_lq[ch.p0:ch.p1 - 1])
        assert ch.rows <= 1024 + 128 or ch.p1 - ch.p0 == 1
        covered += list(range(ch.p0, ch.p1))
    assert covered == list(range(20))
    # pooling segments: every protein's 32-row groups form a contiguous, ordered range inside its segment
    pk2 = PackedProteins.pack([p["seq"] for p in prots], max_rows=1024, max_segment_groups=64)
    assert len(pk2.segments) > 1 and [s.p0 for s in pk2.segments][0] == 0 and pk2.segments[-1].p1 == 20
    for sg in pk2.segments:
        off = pk2.grp_off[sg.grp_off_pos:sg.grp_off_pos + (sg.p1 - sg.p0) + 1]
        assert off[0] == 0 and off[-1] == sg.groups and np.all(np.diff(off) > 0) and sg.groups <= 64
        assert np.all(np.diff(off)[:-1] * 16 >= (pk2.Lq if pk2.order is None else pk2.Lq[pk2.order])[sg.p0:sg.p1 - 1])
    assert sum(c.rows // 16 for c in pk2.chunks) == sum(s.groups for s in pk2.segments)
    with pytest.raises(ValueError, match="does not spell"):
        PackedProteins.pack(["ACD"], [prots[0]["coords"]], ["AC-"], ["ACD"])


def test_prediction_rows_format():
    from mDeepFRI.batch import prediction_rows
    rows = prediction_rows(["q1"], np.array([[0.25, 0.5]], dtype=np.float32))
    assert rows == [["q1", "gcn", 0.25, 0.5]]          # reference pipeline.py:318


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import, include, link or dlopen it
    (mentions in comments/docstrings that say what a kernel is checked against are fine)."""
    forbidden = [r"^\s*(from|import)\s+(oracle|cmap_oracle|gcn_oracle|build_ref)\b", r"#\s*include\s*[\"<][^\">]*oracle",
                 r"libcmap_oracle", r"CDLL\([^)]*oracle", r"dlopen\([^)]*oracle", r"sys\.path[^\n]*oracle",
                 # nor the test / benchmark support package (synthetic workloads, the ONNX exporter): it lives outside the product
                 r"^\s*(from|import)\s+mdfri_testkit\b", r"^\s*from\s+\.\s+import\s+[^\n]*\b(synthetic|onnx_writer)\b"]
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                for pat in forbidden:
                    assert not re.search(pat, txt, flags=re.M), (os.path.join(dirpath, f), pat)


def test_synthetic_generators_are_deterministic_and_pipeline_realistic():
    a = synthetic.synthetic_proteins(seed=5, count=3, length=256, indel_rate=0.05)
    b = synthetic.synthetic_proteins(seed=5, count=3, length=256, indel_rate=0.05)
    assert all(x["seq"] == y["seq"] and np.array_equal(x["coords"], y["coords"]) and x["q_aln"] == y["q_aln"] for x, y in zip(a, b))
    for p in a:
        assert p["q_aln"].replace("-", "") == p["seq"] and len(p["q_aln"]) == len(p["t_aln"])
        assert p["coords"].shape == (len(p["t_aln"].replace("-", "")), 3) and p["coords"].dtype == np.float32
        step = np.linalg.norm(np.diff(p["coords"], axis=0), axis=1)
        assert np.allclose(step, 3.8, atol=0.01)


def test_helix_bundle_generator_is_protein_like():
    """a second coordinate generator with ~5-9 contacts per residue at 6 A (a random walk gives ~12.6)."""
    a = synthetic.synthetic_proteins(seed=7, count=3, length=300, indel_rate=0.05, coords="helix")
    b = synthetic.synthetic_proteins(seed=7, count=3, length=300, indel_rate=0.05, coords="helix")
    assert all(np.array_equal(x["coords"], y["coords"]) for x, y in zip(a, b))
    for p in a:
        xyz = p["coords"].astype(np.float64)
        assert p["coords"].dtype == np.float32 and xyz.shape == (len(p["t_aln"].replace("-", "")), 3)
        assert np.allclose(np.linalg.norm(np.diff(xyz, axis=0), axis=1), 3.8, atol=0.05)
        per_row = (((xyz[:, None] - xyz[None]) ** 2).sum(-1) < 36.0).sum(1)
        assert 6.0 < per_row.mean() < 10.5, per_row.mean()
    def mean_entries(prots):
        return np.mean([(((x[:, None] - x[None]) ** 2).sum(-1) < 36.0).sum(1).mean() for x in (p["coords"].astype(np.float64) for p in prots)])
    assert mean_entries(synthetic.synthetic_proteins(seed=7, count=8, length=300)) > mean_entries(a) + 2.0


def test_insert_gaps_reference_kats_and_contract():
    # reference mDeepFRI/tests/test_alignment.py:38-45
    from mDeepFRI.alignment import AlignmentResult, insert_gaps
    assert insert_gaps("AACT", "AAT", "MMDM") == ("AACT", "AA-T")
    assert insert_gaps("AAT", "AATC", "MMMI") == ("AAT-", "AATC")
    assert insert_gaps("AAT", "FGTC", "XXMI") == ("AAT-", "FGTC")
    assert insert_gaps("", "", "") == ("", "") and insert_gaps("AC", "AC", "") == ("AC", "AC")

    def ref(seq, tgt, aln):  # the semantics of list.insert on growing lists, including indices past the end
        a, b = list(seq), list(tgt)
        for i, c in enumerate(aln):
            if c == "I":
                a.insert(i, "-")
            elif c == "D":
                b.insert(i, "-")
        return "".join(a), "".join(b)

    rng = np.random.default_rng(0)
    for _ in range(200):
        seq = synthetic.random_sequence(rng, int(rng.integers(0, 12)))
        tgt = synthetic.random_sequence(rng, int(rng.integers(0, 12)))
        aln = "".join(rng.choice(list("MIDX"), size=int(rng.integers(0, 20))))
        assert insert_gaps(seq, tgt, aln) == ref(seq, tgt, aln)
    r = AlignmentResult("q", "AACT", "t.pdb", "AAT", "MMDM", 0.9, 0.8, 0.7, "pdb100", coords=None)
    assert (r.gapped_sequence, r.gapped_target, r.coords) == ("AACT", "AA-T", None)
    from mDeepFRI.batch import PackedProteins
    rs = [AlignmentResult("q1", "AACT", "t1", "AAT", "MMDM", coords=np.zeros((3, 3), np.float32)),
          AlignmentResult("q2", "AAT", "t2", "AATC", "MMMI", coords=None),
          AlignmentResult("q3", "AAT", "t3", "AATC", "MMMI", coords=np.zeros((4, 3), np.float32))]
    pk, keep = PackedProteins.from_alignments(rs)
    assert keep == [0, 2] and pk.seqs == ["AACT", "AAT"] and list(pk.Lq) == [4, 3] and list(pk.coord_off) == [0, 3, 7]


def test_model_load_rejects_corrupt_containers(tmp_path):
    """mdf_model_load treats a .mdfw as untrusted input: wrapped element counts, unaligned or out-of-file offsets, zero or
    oversized dimensions and bad ranks must give MDF_EIO before any pointer into the file is formed (the checks run before
    the device is needed, so this is a CPU test; a VALID file then fails with ENODEVICE here, not with EIO)."""
    import struct
    L = _hip.lib()
    w = synthetic.glorot_gcn_weights(0, 8, embed=64, gc_dims=(256, 256, 256), fc_dim=256)
    good = tmp_path / "good.mdfw"
    weights.save_mdfw(str(good), w)
    raw = bytearray(good.read_bytes())
    entry = struct.Struct("<32sI4QQ")
    names = [entry.unpack_from(raw, 12 + i * entry.size)[0].rstrip(b"\0").decode() for i in range(struct.unpack_from("<I", raw, 8)[0])]

    def patched(name, **kw):
        b = bytearray(raw)
        at = 12 + names.index(name) * entry.size
        nm, ndim, d0, d1, d2, d3, off = entry.unpack_from(b, at)
        f = dict(ndim=ndim, d0=d0, d1=d1, d2=d2, d3=d3, off=off)
        f.update(kw)
        entry.pack_into(b, at, nm, f["ndim"], f["d0"], f["d1"], f["d2"], f["d3"], f["off"])
        return bytes(b)

    cases = {
        "wrapping element count": patched("W_aa", d0=26, d1=2**62),                 # 26 * 2^62 * 4 wraps in u64
        "dimension above INT32_MAX": patched("W_fc", d1=2**31),
        "zero dimension": patched("W_gc2", d1=0),
        "rank 0": patched("b_fc", ndim=0),
        "rank 5": patched("b_fc", ndim=5),
        "unaligned offset": patched("W_out", off=entry.unpack_from(raw, 12 + names.index("W_out") * entry.size)[6] + 2),
        "offset past the end": patched("b_out", off=len(raw) + 4096),
        "offset + size wraps": patched("b_out", off=2**64 - 4),
        "offset inside the directory": patched("W_aa", off=16),
        "truncated file": bytes(raw[:len(raw) // 2]),
        "directory longer than the file": bytes(raw[:8]) + struct.pack("<I", 2**31) + bytes(raw[12:]),
    }
    for what, blob in cases.items():
        p = tmp_path / "bad.mdfw"
        p.write_bytes(blob)
        h = ctypes.c_void_p()
        rc = L.mdf_model_load(str(p).encode(), 0, ctypes.byref(h))
        assert rc == _hip.MDF_EIO, (what, rc, _hip.last_error())
        assert not h.value
    h = ctypes.c_void_p()
    rc = L.mdf_model_load(str(good).encode(), 0, ctypes.byref(h))
    if _hip.device_count() == 0:
        assert rc == _hip.MDF_ENODEVICE, _hip.last_error()
    else:
        assert rc == 0
        L.mdf_model_free(h)


def test_get_residues_coordinates_selects_calpha_of_one_chain():
    """reference bio_utils.py:230-255 on an AtomArray-like object (biotite is not installed: the function only needs the
    per-atom arrays).  Mirrors the reference's tests/test_bio_utils.py:24-31 in kind: default chain, invalid chain."""
    from types import SimpleNamespace
    from mDeepFRI.bio_utils import get_residues_coordinates
    rng = np.random.default_rng(0)
    res = ["MET", "LEU", "LEU", "SER", "ALA", "MSE", "GLY"]
    atoms = []
    for ch in ("A", "B"):
        for k, r in enumerate(res):
            for a in ("N", "CA", "C", "O"):
                atoms.append((ch, a, False, r))
    atoms.append(("A", "CA", True, "CA"))         # a calcium ion: hetero, atom name CA -- must not be taken for a C-alpha
    s = SimpleNamespace(chain_id=np.array([a[0] for a in atoms]), atom_name=np.array([a[1] for a in atoms]),
                        hetero=np.array([a[2] for a in atoms]), res_name=np.array([a[3] for a in atoms]),
                        coord=rng.random((len(atoms), 3)).astype(np.float32))
    with pytest.raises(ValueError, match="is not a known amino acid"):
        get_residues_coordinates(s, substitutions={})
    # the default call applies the module-level table, as the reference always does (bio_utils.py:48-193, :252)
    from mDeepFRI import bio_utils
    assert len(bio_utils.substitutions) == 144 and bio_utils.substitutions["MSE"] == "MET" and bio_utils.substitutions["SEP"] == "SER"
    assert get_residues_coordinates(s)[0] == "MLLSAMG"
    seq, xyz = get_residues_coordinates(s, substitutions={"MSE": "MET"})
    assert seq == "MLLSAMG" and xyz.shape == (7, 3) and xyz.dtype == np.float32
    assert np.array_equal(xyz, s.coord[[1 + 4 * k for k in range(7)]])
    assert get_residues_coordinates(s, "B", {"MSE": "MET"})[0] == "MLLSAMG"
    with pytest.raises(ValueError, match="Chain C not found in structure."):
        get_residues_coordinates(s, "C")


def test_default_chunk_rows_is_one_number_everywhere():
    """include/mdfri.h MDF_DEFAULT_CHUNK_ROWS: the planner and the engine fall back to it, the Python layer asks the library, and bench.py's
    --chunk-rows default and the compiled binding's keyword default state it as a literal -- all the same number, a multiple of 32 768
    (whole rounds of 256 x 256 GEMM tiles on 256 CUs)."""
    import re
    from conftest import ROOT
    from mDeepFRI import _hip
    n = _hip.default_chunk_rows()
    assert _hip.DEFAULT_CHUNK_ROWS == n   # the Python-side constant host-only callers use (sharding.plan_summary: no library load)
    from mDeepFRI import batch
    hdr = open(os.path.join(ROOT, "include", "mdfri.h")).read()
    assert int(re.search(r"#define MDF_DENSE_CHUNK_ROWS (\d+)", hdr).group(1)) == batch.DENSE_CHUNK_ROWS <= n   # the dense-map path's own chunk
    assert int(re.search(r"#define MDF_DEFAULT_CHUNK_ROWS (\d+)", hdr).group(1)) == n and n % 32768 == 0
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--chunk-rows", type=int, default=DEFAULT_CHUNK_ROWS,', bench) and "from mDeepFRI._hip import DEFAULT_CHUNK_ROWS" in bench
    pyx = open(os.path.join(ROOT, "tests", "binding", "predict.pyx")).read()
    assert int(re.search(r"int max_rows = (\d+)\)", pyx).group(1)) == n
    lq = np.full(3000, 512, dtype=np.int32)
    from mDeepFRI.batch import PackedProteins
    pk = PackedProteins.pack(["A" * 512] * 3000)
    assert max(c.rows for c in pk.chunks) == n and len(pk.chunks) == -(-3000 * 512 // n)


def test_host_batch_packs_lists_and_refuses_inconsistent_ones():
    """mDeepFRI.batch.HostBatch: the host lists of one batch flattened for mdf_engine_submit_alignments_host (what the compiled binding's
    BatchEngine.submit builds) -- joined bytes, int32 lengths, one (sum Lt, 3) float32 array; lists of different lengths, a gapped target
    of another length than the gapped query, an empty batch are refused before anything reaches the library."""
    from mDeepFRI.batch import HostBatch
    rng = np.random.default_rng(1)
    seqs, q, t = ["ACD", "MKVLA"], ["AC-D", "MKVLA"], ["ACGD", "MK-LA"]
    coords = [rng.standard_normal((4, 3)), rng.standard_normal((4, 3)).astype(np.float64)]
    hb = HostBatch(seqs, coords, q, t)
    assert hb.B == 2 and hb.seq_bytes == b"ACDMKVLA" and hb.q_bytes == b"AC-DMKVLA" and hb.t_bytes == b"ACGDMK-LA"
    assert hb.Lq.tolist() == [3, 5] and hb.La.tolist() == [4, 5] and hb.Lt.tolist() == [4, 4]
    assert hb.xyz.dtype == np.float32 and hb.xyz.shape == (8, 3) and hb.xyz.flags["C_CONTIGUOUS"]
    assert np.array_equal(hb.xyz[4:], coords[1].astype(np.float32))
    with pytest.raises(ValueError, match="one length"):
        HostBatch(seqs, coords[:1], q, t)
    with pytest.raises(ValueError, match="differ in length"):
        HostBatch(seqs, coords, q, ["ACGD", "MKLA"])
    with pytest.raises(ValueError, match="non-empty"):
        HostBatch([], [], [], [])
