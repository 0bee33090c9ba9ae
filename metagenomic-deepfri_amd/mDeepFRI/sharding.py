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


def gather_filtered(offsets, term_idx, kept, local_index, total: int, dst: int = 0, group=None):
    """Gather the COMPACTED output of the ranks (mDeepFRI.output.filter_scores: per local protein the terms with score >=
    threshold, sorted as results.tsv wants them) instead of the dense (n_local, T) score blocks: a few dozen (term, score)
    pairs per protein leave each GPU, ~100x less than the dense rows.

    offsets (n_local+1) int32, term_idx (nnz) int32, kept (nnz) float32: torch tensors of this rank (CUDA for nccl);
    local_index: global protein index of each local row.  Returns (offsets (total+1) int32, term_idx, kept) in global
    protein order on `dst`, None elsewhere.  Two tiny collectives agree on the padded sizes, then one gather per array."""
    import torch
    import torch.distributed as dist

    dev = offsets.device
    cnt = (offsets[1:] - offsets[:-1]).to(torch.int64)
    gidx = torch.as_tensor(list(local_index), dtype=torch.long, device=dev)

    def assemble(blocks):
        counts = torch.zeros(total, dtype=torch.int64, device=blocks[0][0].device)
        for i, c, _, _ in blocks:
            counts[i] = c
        goff = torch.zeros(total + 1, dtype=torch.int64, device=counts.device)
        torch.cumsum(counts, 0, out=goff[1:])
        n = int(goff[-1].item())
        out_t = torch.empty(n, dtype=torch.int32, device=counts.device)
        out_s = torch.empty(n, dtype=torch.float32, device=counts.device)
        for i, c, t, s in blocks:
            if t.numel() == 0:
                continue
            lo = torch.cumsum(c, 0) - c                                   # start of each protein inside the rank's payload
            dest = torch.repeat_interleave(goff[i] - lo, c) + torch.arange(t.numel(), device=t.device)
            out_t[dest] = t
            out_s[dest] = s
        return goff.to(torch.int32), out_t, out_s

    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return assemble([(gidx, cnt, term_idx, kept)])
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if dist.get_backend(group) == "gloo" and dev.type == "cuda":
        cnt, gidx, term_idx, kept = cnt.cpu(), gidx.cpu(), term_idx.cpu(), kept.cpu()   # gloo has no device gather
        dev = cnt.device
    sizes = torch.zeros((world, 2), dtype=torch.long, device=dev)
    sizes[rank, 0], sizes[rank, 1] = cnt.numel(), term_idx.numel()
    dist.all_reduce(sizes, group=group)
    n_max, z_max = int(sizes[:, 0].max().item()), max(int(sizes[:, 1].max().item()), 1)

    def pad(x, n, fill, dtype):
        out = torch.full((n,), fill, dtype=dtype, device=dev)
        out[:x.numel()] = x
        return out

    payload = [pad(gidx, n_max, -1, torch.long), pad(cnt, n_max, 0, torch.long), pad(term_idx, z_max, 0, torch.int32),
               pad(kept, z_max, 0, torch.float32)]
    got = []
    for x in payload:
        bufs = [torch.empty_like(x) for _ in range(world)] if rank == dst else None
        dist.gather(x, bufs, dst=dst, group=group)
        got.append(bufs)
    if rank != dst:
        return None
    blocks = []
    for r in range(world):
        n_r, z_r = int(sizes[r, 0].item()), int(sizes[r, 1].item())
        blocks.append((got[0][r][:n_r], got[1][r][:n_r], got[2][r][:z_r], got[3][r][:z_r]))
    return assemble(blocks)


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


def predict_sharded_filtered(engine, seqs, coords, q_alns, t_alns, threshold: float = 0.1, modes=None, dst: int = 0, group=None,
                             max_rows: int = 65536):
    """predict_sharded with the output stage in front of the gather: every rank filters its own scores on the GPU
    (`score >= threshold`, descending, reference pipeline.py:696-705) and only the survivors travel.
    Returns {mode: (offsets, term_idx, kept)} in input order on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    from .batch import PackedProteins
    from .output import filter_scores

    n = len(seqs)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = partition_by_cost([len(s) for s in seqs], world)[rank]
    modes = list(modes or engine.predictors.keys())
    res = {}
    out = None
    if mine:
        pk = PackedProteins.pack([seqs[i] for i in mine], [coords[i] for i in mine], [q_alns[i] for i in mine],
                                 [t_alns[i] for i in mine], max_rows=max_rows)
        db = engine.upload(pk)
        out = engine.forward_alignments(db)
        engine.check(db)
    for m in modes:
        if mine:
            off, ti, kept = filter_scores(out[m], threshold=threshold)
        else:
            off = torch.zeros(1, dtype=torch.int32, device=engine.device)
            ti = torch.zeros(0, dtype=torch.int32, device=engine.device)
            kept = torch.zeros(0, dtype=torch.float32, device=engine.device)
        res[m] = gather_filtered(off, ti, kept, mine, total=n, dst=dst, group=group)
    return res if rank == dst else None
