"""Multi-rank data path on real kernels: two ranks (gloo plumbing, both on cuda:0 -- the GPU box has one GPU) run
`sharding.predict_sharded` and rank 0 must hold exactly what a single process computes: everything around the collective
(partition, per-rank engine, gather, order restore).  The tests at the end of the file run the SAME paths over RCCL (backend
"nccl") with one rank per DISTINCT device; they switch themselves on wherever two or more devices are visible and skip on a
one-GPU box."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init(rank, world, port, backend):
    """Process group of a test worker; returns the device ordinal of this rank: 0 for everybody under gloo (one-GPU box), one
    device per rank under nccl (RCCL refuses two ranks on one device)."""
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
        assert dist.get_world_size() == world and dist.get_backend() == "nccl"
        return rank
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return 0


def _worker(rank, world, port, q, backend="gloo"):
    import torch
    import torch.distributed as dist
    from mDeepFRI import batch, sharding
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    d = _init(rank, world, port, backend)
    try:
        prots = synthetic.synthetic_proteins(seed=77, count=40, length=(40, 400), indel_rate=0.05)
        w = synthetic.glorot_gcn_weights(seed=0, n_terms=64)
        eng = batch.HotPathEngine({"mf": Predictor("syn", weights=w, device=d)}, device=d, max_rows=4096)
        res = sharding.predict_sharded(eng, [p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                                       [p["t_aln"] for p in prots], max_rows=4096)
        if rank == 0:
            q.put(res["mf"].cpu().numpy())
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_single_process(backend="gloo"):
    import torch.multiprocessing as mp
    from mDeepFRI import batch
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    sharded = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    prots = synthetic.synthetic_proteins(seed=77, count=40, length=(40, 400), indel_rate=0.05)
    w = synthetic.glorot_gcn_weights(seed=0, n_terms=64)
    eng = batch.HotPathEngine({"mf": Predictor("syn", weights=w, device=0)}, device=0, max_rows=4096)
    pk = batch.PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                                   [p["t_aln"] for p in prots], max_rows=4096)
    single = eng.run_alignments(pk)["mf"]
    assert sharded.shape == single.shape == (40, 64)
    assert np.array_equal(sharded, single)   # bitwise: results do not depend on which rank / batch a protein lands in


def _worker_filtered(rank, world, port, q, backend="gloo"):
    import torch.distributed as dist
    from mDeepFRI import batch, sharding
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    d = _init(rank, world, port, backend)
    try:
        prots = synthetic.synthetic_proteins(seed=78, count=30, length=(40, 300), indel_rate=0.05)
        w = synthetic.glorot_gcn_weights(seed=0, n_terms=200)
        eng = batch.HotPathEngine({"mf": Predictor("syn", weights=w, device=d)}, device=d, max_rows=4096)
        res = sharding.predict_sharded_filtered(eng, [p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                                                [p["t_aln"] for p in prots], threshold=0.1, max_rows=4096)
        if rank == 0:
            q.put([x.cpu().numpy() for x in res["mf"]])
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_two_ranks_filtered_gather_equals_single_process_filter(backend="gloo"):
    """The compacted gather (output stage before the collective) delivers exactly what filtering the full single-process
    score matrix gives."""
    import torch.multiprocessing as mp
    from mDeepFRI import batch
    from mdfri_testkit import synthetic
    from mDeepFRI.output import filter_scores
    from mDeepFRI.predict import Predictor
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_filtered, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    off, terms, scores = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    prots = synthetic.synthetic_proteins(seed=78, count=30, length=(40, 300), indel_rate=0.05)
    w = synthetic.glorot_gcn_weights(seed=0, n_terms=200)
    eng = batch.HotPathEngine({"mf": Predictor("syn", weights=w, device=0)}, device=0, max_rows=4096)
    pk = batch.PackedProteins.pack([p["seq"] for p in prots], [p["coords"] for p in prots], [p["q_aln"] for p in prots],
                                   [p["t_aln"] for p in prots], max_rows=4096)
    db = eng.upload(pk)
    out = eng.forward_alignments(db)
    eng.check(db)
    o1, t1, s1 = (x.cpu().numpy() for x in filter_scores(out["mf"], threshold=0.1))
    assert off[-1] > 0 and np.array_equal(off, o1) and np.array_equal(terms, t1) and np.array_equal(scores, s1)


def _mixed_workload():
    """BASELINE.json configs[3] in miniature: L ~ U{128..1024}, 5 % indels (the generator bench.py uses for the real thing)."""
    from mdfri_testkit import synthetic
    lengths = synthetic.uniform_lengths(46, 512)
    return synthetic.bulk_proteins(46, lengths, range(512), indel_rate=0.05)


def _worker_mixed(rank, world, port, q, backend="gloo"):
    import torch.distributed as dist
    from mDeepFRI import batch, sharding
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    d = _init(rank, world, port, backend)
    try:
        seqs, coords, q_alns, t_alns = _mixed_workload()
        w = synthetic.glorot_gcn_weights(seed=0, n_terms=96)
        eng = batch.HotPathEngine({"mf": Predictor("syn", weights=w, device=d)}, device=d, max_rows=32768)
        res = sharding.predict_sharded(eng, seqs, coords, q_alns, t_alns, max_rows=32768)
        if rank == 0:
            q.put(res["mf"].cpu().numpy())
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_configs3_shape_512_mixed_proteins_two_ranks(backend="gloo"):
    """512 proteins with ragged lengths and gapped alignments, dealt to two ranks by cost: rank 0 ends up with every protein's
    scores in input order, bitwise equal to the single-process result, and a sample agrees with the oracle."""
    import torch.multiprocessing as mp
    import cmap_oracle as orc
    import gcn_oracle
    from mDeepFRI import batch, sharding
    from mdfri_testkit import synthetic
    from mDeepFRI.predict import Predictor
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mixed, args=(r, 2, port, q, backend)) for r in range(2)]
    for p in procs:
        p.start()
    sharded = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    seqs, coords, q_alns, t_alns = _mixed_workload()
    assert min(map(len, seqs)) >= 128 and max(map(len, seqs)) <= 1024 and any("-" in a for a in q_alns) and any("-" in a for a in t_alns)
    shards = sharding.partition_by_cost([len(s) for s in seqs], 2)
    loads = [sum(sharding.protein_cost(len(seqs[i])) for i in sh) for sh in shards]
    assert abs(loads[0] - loads[1]) <= 1024 + 15
    w = synthetic.glorot_gcn_weights(seed=0, n_terms=96)
    eng = batch.HotPathEngine({"mf": Predictor("syn", weights=w, device=0)}, device=0, max_rows=65536)
    single = eng.run_alignments(batch.PackedProteins.pack(seqs, coords, q_alns, t_alns, max_rows=65536))["mf"]
    assert sharded.shape == single.shape == (512, 96) and np.array_equal(sharded, single)
    for i in (0, 200, 511):
        cm = orc.build_align_contact_map(coords[i], q_alns[i], t_alns[i], 6.0, 2)
        assert np.max(np.abs(sharded[i] - gcn_oracle.gcn_forward(w, seqs[i], cm))) < 1e-4


def _worker_rccl_one_rank(q):
    """One-rank RCCL group on the one GPU of the box (two ranks cannot share a device: 'Duplicate GPU detected'), with the plans
    told to issue their collectives anyway: the actual RCCL calls of the multi-GPU path -- all_reduce of the shard sizes, gather of
    int64 indices, float32 payloads and int32 terms into device buffers, barrier -- run for real."""
    import torch
    import torch.distributed as dist
    from mDeepFRI import sharding
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        g = torch.Generator(device="cpu").manual_seed(1)
        block = torch.rand((37, 11), generator=g).to(dev)
        order = torch.randperm(37, generator=g).tolist()
        plan = sharding.DenseGatherPlan(37, 11, order, 37, dev, collectives_for_one_rank=True)
        assert not plan.single and plan.world == 1
        for k in (1, 2):
            out = plan.run(block * k)
            torch.cuda.synchronize()
            assert torch.equal(out[order], block * k)
        off = torch.tensor([0, 2, 2, 5], dtype=torch.int32, device=dev)
        ti = torch.tensor([5, 1, 9, 3, 4], dtype=torch.int32, device=dev)
        sc = torch.tensor([0.9, 0.5, 0.3, 0.2, 0.15], device=dev)
        fplan = sharding.FilteredGatherPlan([2, 0, 1], 3, dev, collectives_for_one_rank=True)
        for flag in (True, False):
            goff, gt, gs = fplan.run(off, ti, sc, sizes_may_change=flag)
            assert goff.tolist() == [0, 0, 3, 5] and gt.tolist() == [9, 3, 4, 5, 1] and gs.is_cuda
        dist.barrier()
        q.put("ok")
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_of_the_gather_plans_on_one_rank():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl_one_rank, args=(q,))
    p.start()
    assert q.get(timeout=600) == "ok"
    p.join(timeout=120)
    assert p.exitcode == 0


# ---- the same paths over RCCL, one rank per distinct device: on wherever >= 2 devices are visible -----------------------------------
def _n_devices() -> int:
    import torch
    return torch.cuda.device_count()    # (a count only: does not initialise the GPU in this process)


multi_gpu = pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 HIP devices (one RCCL rank per device)")


@multi_gpu
def test_rccl_two_devices_dense_gather_equals_single_process():
    """sharding.predict_sharded with backend nccl, ranks on cuda:0 and cuda:1: the (B, T) block rank 0 gathers over xGMI is bitwise the
    single-process result."""
    test_two_ranks_equal_single_process(backend="nccl")


@multi_gpu
def test_rccl_two_devices_filtered_gather_equals_single_process_filter():
    test_two_ranks_filtered_gather_equals_single_process_filter(backend="nccl")


@multi_gpu
def test_rccl_two_devices_configs3_shape():
    test_configs3_shape_512_mixed_proteins_two_ranks(backend="nccl")
