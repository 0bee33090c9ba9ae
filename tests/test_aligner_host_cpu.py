"""Host-side logic of the aligner entry points (no device needed): launch plans, the score-mode orientation rule, argument checks that
run before the device is touched, and that the batched aligner fails loudly -- never falls back -- without a GPU."""
import ctypes

import numpy as np
import pytest

from mDeepFRI import _hip
from mDeepFRI.alignment import ScoringMatrix, align_queries_arrays


def _sym(seed=0):
    rng = np.random.default_rng(seed)
    m = rng.integers(-6, 4, size=(24, 24))
    m = ((m + m.T) // 2).astype(np.int32)
    np.fill_diagonal(m, rng.integers(5, 13, size=24))
    return m


def test_plan_offsets_cover_both_trace_formats():
    L = _hip.lib()
    seq_len = np.array([100, 300, 64, 129, 1], dtype=np.int32)
    pq, pt = np.array([0, 1, 2, 4], dtype=np.int32), np.array([1, 0, 3, 4], dtype=np.int32)
    bo, to, oo = (np.zeros(5, dtype=np.int64) for _ in range(3))
    assert L.mdf_nw_plan(_hip.ptr(seq_len), _hip.ptr(pq), _hip.ptr(pt), 4, _hip.ptr(bo), _hip.ptr(to), _hip.ptr(oo)) == 0
    assert bo.tolist() == [0, 200, 800, 928, 930] and oo.tolist() == [0, 400, 800, 993, 995]
    steps32 = lambda lq: (lq + 63 + 3) // 4 * 4        # noqa: E731  one 64-column strip of the 32-bit kernel
    steps16 = lambda lq: (lq + 129 + 3) // 4 * 4       # noqa: E731  one pair of strips of the packed kernel
    want = [max(-(-lt // 64) * steps32(lq), -(-lt // 128) * steps16(lq)) * 64 for lq, lt in ((100, 300), (300, 100), (64, 129), (1, 1))]
    assert np.diff(to).tolist() == want
    assert L.mdf_nw_plan(_hip.ptr(seq_len), _hip.ptr(pq), _hip.ptr(pt), -1, None, None, None) == _hip.MDF_EINVAL


def test_orientation_rule_on_the_host():
    """A pair is turned when the other orientation takes fewer steps: ceil(cols / 128) strip pairs of rows + 129 steps each (rows + 65 for a
    last strip pair whose high strip is empty)."""
    L = _hip.lib()
    seq_len = np.array([100, 300, 128, 129, 500, 40], dtype=np.int32)
    cases = [(0, 1, True),     # 100 x 300: 2 x 232 + 168 steps as given, 1 x 432 turned
             (1, 0, False),    # already the cheaper way round
             (2, 3, False),
             (3, 2, False),
             (5, 4, True),     # 40 x 500: 4 x 172 = 688 as given, 108 turned (one half-filled strip pair)
             (4, 4, False)]
    pq = np.array([c[0] for c in cases], dtype=np.int32)
    pt = np.array([c[1] for c in cases], dtype=np.int32)
    def steps(rows, cols):      # the last strip pair takes rows + 65 steps when the columns end in its low strip
        full, half = (rows + 129 + 3) // 4 * 4, (rows + 65 + 3) // 4 * 4
        n_sp = -(-cols // 128)
        return (n_sp - 1) * full + half if ((cols - 1) & 127) < 64 else n_sp * full
    expect = [steps(seq_len[t], seq_len[q]) < steps(seq_len[q], seq_len[t]) for q, t, _ in cases]
    a, b = pq.copy(), pt.copy()
    m = _sym()
    turned = L.mdf_nw_orient_pairs(_hip.ptr(seq_len), _hip.ptr(a), _hip.ptr(b), len(a), _hip.ptr(m), 24, 10, 1)
    assert turned == sum(expect)
    for k, e in enumerate(expect):
        assert (int(a[k]), int(b[k])) == ((int(pt[k]), int(pq[k])) if e else (int(pq[k]), int(pt[k])))
    assert not expect[1] and expect[4]
    asym = m.copy()
    asym[0, 1] += 1
    a, b = pq.copy(), pt.copy()
    assert L.mdf_nw_orient_pairs(_hip.ptr(seq_len), _hip.ptr(a), _hip.ptr(b), len(a), _hip.ptr(asym), 24, 10, 1) == 0
    assert np.array_equal(a, pq) and np.array_equal(b, pt)
    assert L.mdf_nw_orient_pairs(None, _hip.ptr(a), _hip.ptr(b), len(a), _hip.ptr(m), 24, 10, 1) == _hip.MDF_EINVAL


def test_stepped_entry_checks_its_arguments_before_the_device():
    L = _hip.lib()
    z = np.zeros(4, dtype=np.int64)
    assert L.mdf_nw_best_hits_begin(None, None, None, None, 0, None, 0, None, None, None, 0, 10, 1, 0, b"A", 1 << 20, 0) == _hip.MDF_EINVAL
    assert "NULL workspace" in _hip.last_error()
    assert L.mdf_nw_best_hits_align(None, _hip.ptr(z)) == _hip.MDF_EINVAL
    assert L.mdf_nw_best_hits_finish(None, None, None, None, None, None, None, None, None, 0, None, None) == _hip.MDF_EINVAL
    assert L.mdf_nw_best_hits_abandon(None) == _hip.MDF_EINVAL
    L.mdf_nw_workspace_free(None)                      # harmless


def test_batched_aligner_has_no_cpu_fallback():
    if _hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    assert _hip.current_device() == -1
    h = ctypes.c_void_p()
    assert _hip.lib().mdf_nw_workspace_create(0, None, ctypes.byref(h)) == _hip.MDF_ENODEVICE and not h
    with pytest.raises(_hip.MdfriError, match="no HIP device"):
        align_queries_arrays(["q"], ["ACD"], [{"t": "ACDE"}], scoring_matrix=ScoringMatrix.simple())
    assert len(align_queries_arrays([], [], [], scoring_matrix=ScoringMatrix.simple())) == 0       # nothing to do: no device needed
    with pytest.raises(ValueError, match="at least one candidate"):
        align_queries_arrays(["q"], ["ACD"], [{}], scoring_matrix=ScoringMatrix.simple())
