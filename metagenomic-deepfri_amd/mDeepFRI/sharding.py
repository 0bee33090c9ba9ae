"""Multi-GPU sharding of the hot path: one process per GPU, proteins dealt to ranks by cost, no communication
while computing, ONE gather of the (n_local, T) score blocks to rank 0 at the end (RCCL over xGMI when the
backend is "nccl"; "gloo" on CPU for tests).  The reference has no distributed path at all (SURVEY.md section 2,
"Parallelism census"); every protein is independent through contact map + GCN, weights are replicated.
"""
from __future__ import annotations

import numpy as np


def protein_cost(length: int) -> int:
    """Work of one protein ~ its padded residue rows (the H.W GEMMs dominate and are linear in rows)."""
    return (int(length) + 31) // 32 * 32


def partition_by_cost(lengths, world_size: int):
    """Greedy longest-processing-time assignment.  Returns `world_size` index lists (each sorted by length, as the
    reference sorts its work list, pipeline.py:529-533).  Deterministic."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += protein_cost(lengths[i])
    for s in shards:
        s.sort(key=lambda i: (int(lengths[i]), i))
    return shards


def gather_scores(local_scores, local_index, total: int, dst: int = 0, group=None):
    """Gather per-rank score blocks to `dst` and restore the original protein order.

    local_scores: torch tensor (n_local, T) on this rank's device (CUDA for nccl, CPU for gloo);
    local_index:  the global protein indices of its rows (sequence of ints);  total: global protein count.
    Returns the (total, T) tensor on `dst`, None elsewhere.  Shards are padded to the largest shard so that a single
    fixed-size gather is issued (7 peers write into rank 0 over 7 distinct xGMI links in parallel)."""
    import torch
    import torch.distributed as dist

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        out = torch.empty((total, local_scores.shape[1]), dtype=local_scores.dtype, device=local_scores.device)
        out[torch.as_tensor(list(local_index), dtype=torch.long, device=local_scores.device)] = local_scores
        return out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and local_scores.is_cuda:
        local_scores = local_scores.cpu()  # gloo has no device gather; RCCL ("nccl") gathers in HBM over xGMI
    dev = local_scores.device
    n_local, T = local_scores.shape
    counts = torch.zeros(world, dtype=torch.long, device=dev)
    counts[rank] = n_local
    dist.all_reduce(counts, group=group)          # tiny: agree on shard sizes
    n_max = int(counts.max().item())
    payload = torch.zeros((n_max, T), dtype=local_scores.dtype, device=dev)
    payload[:n_local] = local_scores
    idx = torch.full((n_max,), -1, dtype=torch.long, device=dev)
    idx[:n_local] = torch.as_tensor(list(local_index), dtype=torch.long, device=dev)
    if rank == dst:
        bufs = [torch.empty_like(payload) for _ in range(world)]
        ibufs = [torch.empty_like(idx) for _ in range(world)]
    else:
        bufs = ibufs = None
    dist.gather(payload, bufs, dst=dst, group=group)
    dist.gather(idx, ibufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty((total, T), dtype=local_scores.dtype, device=dev)
    for b, i in zip(bufs, ibufs):
        keep = i >= 0
        out[i[keep]] = b[keep]
    return out


def predict_sharded(engine, seqs, coords, q_alns, t_alns, modes=None, dst: int = 0, group=None, max_rows: int = 65536):
    """Whole multi-GPU path for one workload known to every rank: deal proteins to ranks by cost, run the fused hot path
    on this rank's shard with `engine` (a mDeepFRI.batch.HotPathEngine bound to this rank's GPU), gather once.
    Returns {mode: (N, T) torch tensor in input order} on `dst`, None on the other ranks.  With an uninitialised
    process group it degenerates to the single-GPU path."""
    import torch
    import torch.distributed as dist
    from .batch import PackedProteins

    n = len(seqs)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = partition_by_cost([len(s) for s in seqs], world)[rank]
    modes = list(modes or engine.predictors.keys())
    widths = [engine.predictors[m].n_terms for m in modes]
    if mine:
        pk = PackedProteins.pack([seqs[i] for i in mine], [coords[i] for i in mine], [q_alns[i] for i in mine],
                                 [t_alns[i] for i in mine], max_rows=max_rows)
        db = engine.upload(pk)
        out = engine.forward_alignments(db)
        engine.check(db)
        block = torch.cat([out[m] for m in modes], dim=1)
    else:
        block = torch.zeros((0, sum(widths)), dtype=torch.float32, device=engine.device)
    full = gather_scores(block, mine, total=n, dst=dst, group=group)
    if full is None:
        return None
    res, c = {}, 0
    for m, w in zip(modes, widths):
        res[m] = full[:, c:c + w]
        c += w
    return res
