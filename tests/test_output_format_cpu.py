"""The results.tsv formatter of the library (`mdf_results_format_host`, host code: runs without a GPU) against the reference's
text round trip (oracle/output_oracle.py restates pipeline.py:684-716) and against Python's own `f"{score:.4f}"`."""
import numpy as np
import pytest

import output_oracle
from mDeepFRI.output import results_rows, results_text


def _filter_np(s, thr=0.1):
    """numpy stand-in for the device filter (tests/test_gpu_output.py checks the real one): offsets / term indices / scores."""
    off, ti, kept = [0], [], []
    for row in s:
        keep = sorted([(i, float(v)) for i, v in enumerate(row) if float(v) >= thr], key=lambda iv: iv[1], reverse=True)
        ti += [i for i, _ in keep]
        kept += [v for _, v in keep]
        off.append(len(ti))
    return np.array(off, np.int32), np.array(ti, np.int32), np.array(kept, np.float32)


def test_lines_equal_the_reference_text_round_trip():
    rng = np.random.default_rng(5)
    B, T = 40, 320
    s = (rng.random((B, T)) ** 4).astype(np.float32)
    ids = [f"prot_{i}|x" for i in range(B)]
    terms = [f"GO:{1000000 + i:07d}" for i in range(T)]
    names = [f"name of term {i}, with a comma" if i % 7 else "β-glucosidase activity" for i in range(T - 3)]      # three terms without a name
    aln = {"prot_3|x": ["True", "1abc_A", "pdb100", "0.87", "0.91", "0.78"], "prot_9|x": [True, "AF-P1", "afdb", 0.5, 1.0, np.float32(0.25)]}
    off, ti, kept = _filter_np(s)
    for data in (aln, None):
        got = results_rows(ids, "gcn", "Cellular Component", terms, names, off, ti, kept, data)
        exp = output_oracle.results_lines(output_oracle.matrix_text(ids, "gcn", terms, s), "Cellular Component", names, data)
        assert got == exp and len(got) > B
    assert results_text(ids, "gcn", "x", terms, names, off, ti, kept).decode("utf-8") == "".join(results_rows(ids, "gcn", "x", terms, names, off, ti, kept))


def test_score_text_is_pythons_for_every_kind_of_float():
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.random(20000).astype(np.float32), (rng.random(5000) * 1e-4).astype(np.float32), (rng.random(5000) * 1e5).astype(np.float32),
        rng.standard_normal(5000).astype(np.float32) * np.float32(3e9), np.frombuffer(rng.bytes(40000), dtype=np.float32),
        np.array([0.0, -0.0, 1.0, 0.1, 0.5, 0.99995, 0.99994999, 0.00005, 0.00015, 0.00025, 0.12345, 0.12355, 9.9999e8, 1e9, 1.00000001e9, 3.4e38,
                  1e-45, np.inf, -np.inf, np.nan, 0.0625 + 2.0 ** -20, 123456.78125, 999999.99], dtype=np.float32)])
    n = len(vals)
    txt = results_text(["q"], "n", "m", ["t"], ["g"], np.array([0, n], np.int32), np.zeros(n, np.int32), vals).decode()
    got = [line.split("\t")[4] for line in txt.splitlines()]
    assert got == [f"{float(v):.4f}" for v in vals]


def test_empty_and_argument_checks():
    z = np.zeros(0, np.int32)
    assert results_text([], "gcn", "m", ["t"], ["g"], np.zeros(1, np.int32), z, np.zeros(0, np.float32)) == b""
    assert results_rows(["a", "b"], "gcn", "m", ["t"], ["g"], np.zeros(3, np.int32), z, np.zeros(0, np.float32)) == []
    with pytest.raises(ValueError, match="term index 1 out of range"):
        results_text(["a"], "gcn", "m", ["t"], ["g"], np.array([0, 1], np.int32), np.array([1], np.int32), np.ones(1, np.float32))
    with pytest.raises(ValueError, match="one entry per query"):
        results_text(["a"], "gcn", "m", ["t"], ["g"], np.array([0, 1, 1], np.int32), np.array([0], np.int32), np.ones(1, np.float32))


def test_prediction_matrix_text_is_the_csv_writers():
    """`prediction_matrix_text` (library: `mdf_matrix_format_host`) against the text the reference's csv.writer produces
    (oracle/output_oracle.py::matrix_text restates pipeline.py:566-571, 318-319): floats of every magnitude, ids that need quoting."""
    from mDeepFRI.output import prediction_matrix_text
    rng = np.random.default_rng(1)
    B, T = 30, 257
    with np.errstate(all="ignore"):
        s = np.concatenate([rng.random((10, T)), rng.random((5, T)) * 1e-6, np.exp(rng.standard_normal((5, T)) * 30),
                            -rng.random((5, T)) * 1e17, np.frombuffer(rng.bytes(5 * T * 4), dtype=np.float32).reshape(5, T).astype(np.float64)]).astype(np.float32)
    s[0, :12] = [0.0, -0.0, 1.0, 0.1, 1e-4, 9.9999e-5, 1e16, 9.999999e15, 1e-45, 3.4028235e38, 123456.0, 0.5]
    s[1, :3] = [np.inf, -np.inf, np.nan]
    ids = [f"prot_{i}" for i in range(B)]
    terms = [f"GO:{i:07d}" for i in range(T)]
    with np.errstate(all="ignore"):
        assert prediction_matrix_text(ids, s, "gcn", terms).decode() == output_oracle.matrix_text(ids, "gcn", terms, s)
        assert prediction_matrix_text(ids, s, "cnn").decode() == output_oracle.matrix_text(ids, "cnn", terms, s).split("\r\n", 1)[1]
    odd = ['with\ttab', 'with "quote"', "multi\nline", "", "plain"]
    assert prediction_matrix_text(odd, s[:5], "gcn", terms).decode() == output_oracle.matrix_text(odd, "gcn", terms, s[:5])
    assert prediction_matrix_text([], np.zeros((0, T), np.float32), "gcn") == b""
    with pytest.raises(ValueError, match="scores must be"):
        prediction_matrix_text(ids, s[:3])


def test_repr_text_for_many_random_floats():
    from mDeepFRI.output import prediction_matrix_text
    rng = np.random.default_rng(2)
    v = np.concatenate([np.frombuffer(rng.bytes(400000), dtype=np.float32), rng.random(100000).astype(np.float32),
                        (10.0 ** rng.uniform(-45, 38, 50000)).astype(np.float32)]).reshape(1, -1)
    got = prediction_matrix_text(["q"], v, "n").decode().rstrip("\r\n").split("\t")[2:]
    assert got == [repr(float(x)) for x in v[0]]


def test_matrix_format_entry_capacity_paths():
    """`mdf_matrix_format_host` straight through ctypes: the sizing call, a buffer of exactly the needed size and a worst-case buffer give
    the same bytes, for 1 and several threads; `write_prediction_matrix` writes header + the same rows."""
    import ctypes
    from mDeepFRI import _hip
    from mDeepFRI.output import _concat
    rng = np.random.default_rng(3)
    B, T = 300, 400
    s = (rng.random((B, T)) ** 6).astype(np.float32)
    pre_b, pre_off = _concat([f"id{i}\tgcn" for i in range(B)])
    n = ctypes.c_int64()
    L = _hip.lib()
    assert L.mdf_matrix_format_host(pre_b, _hip.ptr(pre_off), _hip.ptr(s), B, T, None, 0, 0, ctypes.byref(n)) == _hip.MDF_ECAPACITY
    need = n.value
    texts = []
    for cap, threads in ((need, 1), (need, 5), (int(pre_off[-1]) + B * (T * 25 + 2), 1), (int(pre_off[-1]) + B * (T * 25 + 2), 7)):
        out = np.zeros(cap, dtype=np.uint8)
        assert L.mdf_matrix_format_host(pre_b, _hip.ptr(pre_off), _hip.ptr(s), B, T, _hip.ptr(out), cap, threads, ctypes.byref(n)) == 0 and n.value == need
        texts.append(out[:need].tobytes())
    assert len(set(texts)) == 1 and texts[0].decode() == output_oracle.matrix_text([f"id{i}" for i in range(B)], "gcn", [], s).split("\r\n", 1)[1]
    out = np.zeros(need - 1, dtype=np.uint8)
    assert L.mdf_matrix_format_host(pre_b, _hip.ptr(pre_off), _hip.ptr(s), B, T, _hip.ptr(out), need - 1, 2, ctypes.byref(n)) == _hip.MDF_ECAPACITY and n.value == need
    import io
    from mDeepFRI.output import write_prediction_matrix
    fh = io.BytesIO()
    terms = [f"t{k}" for k in range(T)]
    assert write_prediction_matrix(fh, [f"id{i}" for i in range(B)], s, "gcn", terms) == len(fh.getvalue())
    assert fh.getvalue().decode() == output_oracle.matrix_text([f"id{i}" for i in range(B)], "gcn", terms, s)
