"""Multi-GPU path on CPU: cost-balanced partition and the final gather, world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mDeepFRI import sharding


def test_partition_covers_everything_and_balances():
    rng = np.random.default_rng(4)
    L = rng.integers(128, 1025, size=1000)
    shards = sharding.partition_by_cost(L, 8)
    flat = sorted(i for s in shards for i in s)
    assert flat == list(range(1000))
    loads = [sum(sharding.protein_cost(L[i]) for i in s) for s in shards]
    assert max(loads) - min(loads) <= 1024 + 15                     # LPT bound: at most one longest item apart
    for s in shards:
        assert [L[i] for i in s] == sorted(L[i] for i in s)        # each shard sorted by length (pipeline.py:529)
    assert sharding.partition_by_cost(L, 8) == shards               # deterministic
    assert sharding.partition_by_cost([5, 5], 4) == [[0], [1], [], []]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_dry_plan_of_the_sharded_baseline_configs_is_balanced(world):
    """BASELINE configs[3] (100 000 proteins, L ~ U{128..1024}) and configs[4] (500 000, length histogram): predicted
    imbalance of the padded rows below 1 % at every rank count the driver will launch (`bench.py --gpus N --dry-plan`)."""
    from mdfri_testkit import synthetic
    for lengths in (synthetic.uniform_lengths(46, 100_000), synthetic.histogram_lengths(47, 500_000)):   # bench.py's seeds
        plan = sharding.plan_summary(lengths, world)
        assert sum(plan["proteins"]) == len(lengths) and plan["world"] == world
        assert plan["imbalance"] < 0.01, plan


def test_gather_single_process_restores_order():
    scores = torch.arange(12, dtype=torch.float32).reshape(4, 3)
    out = sharding.gather_scores(scores, [2, 0, 3, 1], total=4)
    assert torch.equal(out[[2, 0, 3, 1]], scores)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lengths, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shards = sharding.partition_by_cost(lengths, world)
        mine = shards[rank]
        # stand-in for the per-rank hot path: row i holds values derived from the global protein index
        local = torch.stack([torch.full((T,), float(i)) + torch.arange(T) / 100.0 for i in mine]) if mine else torch.zeros((0, T))
        out = sharding.gather_scores(local, mine, total=len(lengths), dst=0)
        # the plan form bench.py keeps across steps: several runs with different payloads through ONE plan
        plan = sharding.DenseGatherPlan(len(mine), T, mine, len(lengths), "cpu", dst=0)
        for k in (1, 2, 3):
            again = plan.run(local * k)
            assert (again is None) == (rank != 0)
            if rank == 0:
                assert torch.equal(again, out * k)
        if rank == 0:
            q.put(out.numpy())
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [7, 64])
def test_gather_world_size_2_gloo(n):
    lengths = list(np.random.default_rng(n).integers(30, 900, size=n))
    T = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, lengths, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = np.stack([np.full(T, float(i)) + np.arange(T) / 100.0 for i in range(n)]).astype(np.float32)
    assert np.array_equal(out, exp)


def _csr_of(i, T):
    """deterministic fake output of protein i: (terms, scores)"""
    k = (i * 7) % 5            # 0..4 kept terms (some proteins keep none)
    terms = [(i + 3 * j) % T for j in range(k)]
    return terms, [1.0 - 0.1 * j - i / 1000.0 for j in range(k)]


def _worker_csr(rank, world, port, n, T, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = sharding.partition_by_cost([50 + (i * 37) % 400 for i in range(n)], world)[rank]
        off, terms, scores = [0], [], []
        for i in mine:
            t, s = _csr_of(i, T)
            terms += t
            scores += s
            off.append(len(terms))
        out = sharding.gather_filtered(torch.tensor(off, dtype=torch.int32), torch.tensor(terms, dtype=torch.int32),
                                       torch.tensor(scores, dtype=torch.float32), mine, total=n, dst=0)
        plan = sharding.FilteredGatherPlan(mine, n, "cpu", dst=0)
        for _ in range(3):   # re-used plan: padded buffers sized on the first run
            again = plan.run(torch.tensor(off, dtype=torch.int32), torch.tensor(terms, dtype=torch.int32), torch.tensor(scores, dtype=torch.float32))
            assert (again is None) == (rank != 0)
            if rank == 0:
                assert all(torch.equal(a, b) for a, b in zip(again, out))
        # the steady-state form bench.py uses: sizes fixed by the first run, no size agreement afterwards
        for _ in range(2):
            again = plan.run(torch.tensor(off, dtype=torch.int32), torch.tensor(terms, dtype=torch.int32), torch.tensor(scores, dtype=torch.float32),
                             sizes_may_change=False)
            assert (again is None) == (rank != 0)
            if rank == 0:
                assert all(torch.equal(a, b) for a, b in zip(again, out))
        plan.check()
        if n >= 9:
            # rank 1 outgrows the plan: nobody may hang in a collective, and EVERY rank raises (at the latest in check())
            grow = 40 if rank == 1 else 1
            big_t = torch.tensor(terms * grow, dtype=torch.int32)
            big_s = torch.tensor(scores * grow, dtype=torch.float32)
            big_off = torch.tensor([o * grow for o in off], dtype=torch.int32)
            with pytest.raises(RuntimeError, match="exceeded the planned payload"):
                plan.run(big_off, big_t, big_s, sizes_may_change=False)
                plan.check()
            # the plan recovers: the next run re-agrees on the sizes
            again = plan.run(torch.tensor(off, dtype=torch.int32), torch.tensor(terms, dtype=torch.int32), torch.tensor(scores, dtype=torch.float32),
                             sizes_may_change=False)
            if rank == 0:
                assert all(torch.equal(a, b) for a, b in zip(again, out))
            plan.check()
        if rank == 0:
            q.put([x.numpy() for x in out])
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [1, 9, 50])
def test_gather_filtered_world_size_2_gloo(n):
    T = 11
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_csr, args=(r, 2, port, n, T, q)) for r in range(2)]
    for p in procs:
        p.start()
    off, terms, scores = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert off.shape == (n + 1,) and off[0] == 0
    for i in range(n):
        t, s = _csr_of(i, T)
        assert list(terms[off[i]:off[i + 1]]) == t
        assert np.array_equal(scores[off[i]:off[i + 1]], np.asarray(s, dtype=np.float32))


def _worker_w8(rank, world, port, n, T, empty_rank, q):
    """Both gather plans in one group of `world` ranks: ranks that own no protein at all (n < world) and a rank whose filter keeps
    nothing still take part in every collective with empty payloads."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lengths = [50 + (i * 37) % 400 for i in range(n)]
        mine = sharding.partition_by_cost(lengths, world)[rank]
        local = torch.stack([torch.full((T,), float(i)) + torch.arange(T) / 100.0 for i in mine]) if mine else torch.zeros((0, T))
        dense = sharding.DenseGatherPlan(len(mine), T, mine, n, "cpu", dst=0)
        # (the plan returns ITS output buffer: rank 0 copies it before the next run)
        first_dense = dense.run(local)
        first_dense = first_dense.clone() if rank == 0 else None
        second_dense = dense.run(local * 2)
        off, terms, scores = [0], [], []
        for i in mine:
            t, s = ([], []) if rank == empty_rank else _csr_of(i, T)
            terms += t
            scores += s
            off.append(len(terms))
        plan = sharding.FilteredGatherPlan(mine, n, "cpu", dst=0)
        args = (torch.tensor(off, dtype=torch.int32), torch.tensor(terms, dtype=torch.int32), torch.tensor(scores, dtype=torch.float32))
        first = plan.run(*args)
        again = plan.run(*args, sizes_may_change=False)
        plan.check()
        if rank == 0:
            assert torch.equal(second_dense, first_dense * 2) and all(torch.equal(a, b) for a, b in zip(first, again))
            q.put((first_dense.numpy(), [x.numpy() for x in first]))
        else:
            assert second_dense is None and first is None and again is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,empty_rank", [(5, -1), (8, 0), (50, 3), (50, 0)])
def test_both_gather_plans_world_size_8_gloo(n, empty_rank):
    """SURVEY 8e at the node's real rank count: 8 gloo ranks, (a) fewer proteins than ranks -- three ranks own nothing --, (b) a rank
    (also the destination itself) whose filter keeps nothing.  Dense rows and the compacted (offsets, terms, scores) arrive in input order."""
    world, T = 8, 11
    lengths = [50 + (i * 37) % 400 for i in range(n)]
    shards = sharding.partition_by_cost(lengths, world)
    if n < world:
        assert sum(1 for s in shards if not s) == world - n
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_w8, args=(r, world, port, n, T, empty_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    dense, (off, terms, scores) = q.get(timeout=240)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = np.stack([np.full(T, float(i)) + np.arange(T) / 100.0 for i in range(n)]).astype(np.float32)
    assert np.array_equal(dense, exp)
    emptied = set(shards[empty_rank]) if empty_rank >= 0 else set()
    assert off.shape == (n + 1,) and off[0] == 0
    for i in range(n):
        t, s = ([], []) if i in emptied else _csr_of(i, T)
        assert list(terms[off[i]:off[i + 1]]) == t, i
        assert np.array_equal(scores[off[i]:off[i + 1]], np.asarray(s, dtype=np.float32))


def test_gather_plans_of_configs4_fit_their_index_types():
    """BASELINE configs[4] (500 000 proteins, proteome length histogram, MF + BP + CC = 2 752 score columns) on 8 ranks, plan construction
    only (no payload is allocated): imbalance of the deal below 1 %, the dense gather's element counts (per-rank payload, the destination's
    receive block and output: 5.5 GB) stay below 2^31 elements per collective and are indexed with int64 rows, and the compacted gather's
    int32 offsets hold the survivors of the whole job at the bench's keep rate with a wide margin."""
    from mdfri_testkit import synthetic
    lengths = synthetic.histogram_lengths(47, 500_000)
    world, width = 8, sum(synthetic.GO_TERMS[m] for m in ("mf", "bp", "cc"))
    plan = sharding.plan_summary(lengths, world)
    assert plan["imbalance"] < 0.01 and sum(plan["proteins"]) == 500_000
    fp = sharding.DenseGatherPlan.footprint(plan["proteins"], width)
    assert fp["payload_elems"] < 2 ** 31 and fp["recv_elems"] < 2 ** 31 and fp["out_elems"] < 2 ** 31
    assert 5.4e9 < fp["out_bytes"] < 5.6e9 and fp["row_index_dtype"] == "int64"
    # the compacted form: results.tsv keeps a few dozen terms per protein; even 1 000 per protein would fit the int32 offsets
    assert sharding.FilteredGatherPlan.max_survivors() >= 500_000 * 1000
    # chunk row offsets inside the engine are int32 per CHUNK (<= max_rows rows), never per job
    from mDeepFRI import _hip
    assert max(plan["padded_rows"]) > 2 ** 24 and max(plan["chunks"]) * _hip.default_chunk_rows() >= max(plan["padded_rows"])


def test_gather_filtered_single_process():
    off = torch.tensor([0, 2, 2, 3], dtype=torch.int32)
    t = torch.tensor([5, 1, 9], dtype=torch.int32)
    s = torch.tensor([0.9, 0.5, 0.3])
    goff, gt, gs = sharding.gather_filtered(off, t, s, [2, 0, 1], total=3)
    assert goff.tolist() == [0, 0, 1, 3] and gt.tolist() == [9, 5, 1] and gs.tolist() == pytest.approx([0.3, 0.9, 0.5])


def test_cost_model_uses_the_library_row_alignment():
    """sharding.GROUP_ROWS mirrors MDF_GROUP_ROWS (include/mdfri.h): the cost of a protein is its padded residue rows."""
    from mDeepFRI import _hip, sharding
    assert sharding.GROUP_ROWS == _hip.lib().mdf_group_rows()
    assert sharding.protein_cost(1) == sharding.GROUP_ROWS and sharding.protein_cost(513) == 528
