"""The library's batch planner (mdf_plan_create, csrc/engine.hip) needs no GPU: chunks of consecutive proteins within max_rows,
16-row padding per protein (MDF_GROUP_ROWS), chunk totals rounded to 128, pooling segments within max_segment_groups -- checked here against a
straightforward Python restatement of the layout rules of include/mdfri.h ("Residue-row layout")."""
import numpy as np
import pytest

from mDeepFRI import _hip
from mDeepFRI.batch import PackedProteins


def _reference_plan(Lq, max_rows, max_segment_groups):
    pad = (np.asarray(Lq, dtype=np.int64) + 15) // 16 * 16
    B = len(Lq)
    chunks, row_off, p0 = [], [], 0
    while p0 < B:
        p1, rows = p0, 0
        while p1 < B and (p1 == p0 or rows + pad[p1] <= max_rows):
            rows += pad[p1]
            p1 += 1
        ro = np.concatenate(([0], np.cumsum(pad[p0:p1])))
        ro[-1] = max((ro[-1] + 127) // 128 * 128, 128)
        chunks.append([p0, p1, int(ro[-1]), sum(len(r) for r in row_off)])
        row_off.append(ro)
        p0 = p1
    segments, grp_off, cur, groups = [], [], [], 0

    def close():
        first, last = chunks[cur[0]], chunks[cur[-1]]
        off = []
        for ci in cur:
            ch = chunks[ci]
            off += list(ch[5] + row_off[ci][:-1] // 16)
        segments.append([first[0], last[1], groups, sum(len(g) for g in grp_off)])
        grp_off.append(np.asarray(off + [groups]))

    for ci, ch in enumerate(chunks):
        g = ch[2] // 16
        if cur and groups + g > max_segment_groups:
            close()
            cur, groups = [], 0
        ch += [len(segments), groups]
        cur.append(ci)
        groups += g
    close()
    return chunks, segments, np.concatenate(row_off), np.concatenate(grp_off)


@pytest.mark.parametrize("keep_order", [True, False])
@pytest.mark.parametrize("seed", range(6))
def test_planner_matches_the_layout_rules(seed, keep_order):
    """keep_order: the proteins are visited as given; default: shortest first, stably (mdf_plan_order = the stable argsort of the lengths; the
    packed arrays themselves stay in the caller's order), and the tables are those of the sorted lengths."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 400))
    Lq = rng.integers(1, int(rng.choice([40, 300, 1100])), size=n)
    max_rows = int(rng.choice([128, 1024, 4096, 65536]))
    max_groups = int(rng.choice([8, 64, 1 << 20]))
    pk = PackedProteins.pack(["A" * int(l) for l in Lq], max_rows=max_rows, max_segment_groups=max_groups, keep_order=keep_order)
    assert np.array_equal(pk.Lq, Lq)                                   # the batch is packed as given either way
    if keep_order or np.all(np.diff(Lq) >= 0):
        assert pk.order is None
    else:
        assert np.array_equal(pk.order, np.argsort(Lq, kind="stable"))
        Lq = Lq[pk.order]
    chunks, segments, row_off, grp_off = _reference_plan(Lq, max_rows, max_groups)
    assert [[c.p0, c.p1, c.rows, c.row_off_pos, c.segment, c.group_base] for c in pk.chunks] == chunks
    assert [[s.p0, s.p1, s.groups, s.grp_off_pos] for s in pk.segments] == segments
    assert np.array_equal(pk.chunk_row_off, row_off) and np.array_equal(pk.grp_off, grp_off)
    assert pk.max_chunk_rows == max(c[2] for c in chunks)
    # layout invariants of include/mdfri.h
    for c in pk.chunks:
        ro = pk.chunk_row_off[c.row_off_pos:c.row_off_pos + c.p1 - c.p0 + 1]
        assert (ro[:-1] % 16 == 0).all() and ro[-1] % 128 == 0 and ro[-1] == c.rows
        assert c.p1 - c.p0 == 1 or c.rows <= max(max_rows, 128) + 96


def test_planner_rejects_bad_input():
    L = _hip.lib()
    h = _hip.c_void_p()
    lq = np.array([5, 0, 7], dtype=np.int32)
    assert L.mdf_plan_create(_hip.ptr(lq), 3, 1024, 0, _hip.ctypes.byref(h)) == _hip.MDF_EINVAL
    assert b"empty sequence" in L.mdf_last_error()
    assert L.mdf_plan_create(_hip.ptr(lq), 0, 1024, 0, _hip.ctypes.byref(h)) == _hip.MDF_EINVAL
    with pytest.raises(ValueError, match="empty"):
        PackedProteins.pack([])


def test_engine_entry_points_fail_without_a_device():
    """No CPU fallback: without a GPU the engine cannot even be created."""
    if _hip.device_count() > 0:
        pytest.skip("a GPU is visible")
    L = _hip.lib()
    h = _hip.c_void_p()
    models = (_hip.c_void_p * 1)(None)
    rc = L.mdf_engine_create(models, 1, 0, None, _hip.ctypes.byref(h))
    assert rc == _hip.MDF_ENODEVICE and b"no CPU fallback" in L.mdf_last_error()
