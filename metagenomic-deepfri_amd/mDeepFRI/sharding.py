"""Multi-GPU sharding of the hot path: one process per GPU, proteins dealt to ranks by cost, no communication
while computing, ONE gather of the (n_local, T) score blocks to rank 0 at the end (RCCL over xGMI when the
backend is "nccl"; "gloo" on CPU for tests).  The reference has no distributed path at all (SURVEY.md section 2,
"Parallelism census"); every protein is independent through contact map + GCN, weights are replicated.
"""
from __future__ import annotations

import numpy as np


GROUP_ROWS = 16    # = MDF_GROUP_ROWS of include/mdfri.h (mdf_group_rows()): a protein's residue rows are padded to a multiple of it


def protein_cost(length: int) -> int:
    """Work of one protein ~ its padded residue rows (the H.W GEMMs dominate and are linear in rows)."""
    return (int(length) + GROUP_ROWS - 1) // GROUP_ROWS * GROUP_ROWS


def partition_by_cost(lengths, world_size: int):
    """Greedy longest-processing-time assignment: proteins by decreasing length (ties by index), each to the rank with the
    smallest load so far (ties: lowest rank).  Returns `world_size` index lists (each sorted by length, as the reference sorts
    its work list, pipeline.py:529-533).  Deterministic.  The order is one NumPy lexsort and the deal a binary heap --
    O(N log world), 0.3 s for the 500 000 proteins of BASELINE configs[4] (it runs on every rank before the first launch)."""
    import heapq
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    order = np.lexsort((np.arange(n), -lengths))
    cost = ((lengths + GROUP_ROWS - 1) // GROUP_ROWS * GROUP_ROWS)[order].tolist()
    heap = [(0, r) for r in range(world_size)]          # (load, rank): the heap order IS the tie rule
    owner = np.empty(n, dtype=np.int64)
    own = owner.tolist()
    for k, c in enumerate(cost):
        load, r = heap[0]
        own[k] = r
        heapq.heapreplace(heap, (load + c, r))
    owner = np.asarray(own, dtype=np.int64)
    shards = []
    for r in range(world_size):
        mine = order[owner == r]
        shards.append(mine[np.lexsort((mine, lengths[mine]))].tolist())
    return shards


def plan_summary(lengths, world_size: int, max_rows: int = None):
    """What `bench.py --dry-plan` prints and tests/test_sharding_cpu.py asserts on: per rank the proteins, padded residue rows
    (the cost model) and chunks of `max_rows`, plus the predicted imbalance max/mean - 1 of the padded rows.  CPU only."""
    if not max_rows:
        from ._hip import DEFAULT_CHUNK_ROWS as max_rows   # (a constant of the module: no library, no HIP runtime is loaded for a plan on paper)
    max_rows = int(max_rows)
    lengths = np.asarray(lengths, dtype=np.int64)
    pad = (lengths + GROUP_ROWS - 1) // GROUP_ROWS * GROUP_ROWS
    shards = partition_by_cost(lengths, world_size)
    rows = [int(pad[s].sum()) for s in shards]
    mean = sum(rows) / max(world_size, 1)
    return {"world": world_size, "proteins": [len(s) for s in shards], "padded_rows": rows,
            "chunks": [int(-(-r // max_rows)) for r in rows],
            "imbalance": (max(rows) / mean - 1.0) if mean else 0.0}


def gather_scores(local_scores, local_index, total: int, dst: int = 0, group=None):
    """Gather per-rank score blocks to `dst` and restore the original protein order (one-shot form of DenseGatherPlan).

    local_scores: torch tensor (n_local, T) on this rank's device (CUDA for nccl, CPU for gloo);
    local_index:  the global protein indices of its rows (sequence of ints);  total: global protein count.
    Returns the (total, T) tensor on `dst`, None elsewhere.  Shards are padded to the largest shard so that a single
    fixed-size gather is issued (7 peers write into rank 0 over 7 distinct xGMI links in parallel)."""
    return DenseGatherPlan(local_scores.shape[0], local_scores.shape[1], local_index, total, local_scores.device,
                           dtype=local_scores.dtype, dst=dst, group=group).run(local_scores)


def gather_filtered(offsets, term_idx, kept, local_index, total: int, dst: int = 0, group=None):
    """Gather the COMPACTED output of the ranks (mDeepFRI.output.filter_scores: per local protein the terms with score >=
    threshold, sorted as results.tsv wants them) instead of the dense (n_local, T) score blocks: a few dozen (term, score)
    pairs per protein leave each GPU, ~100x less than the dense rows.  One-shot form of
    FilteredGatherPlan.

    offsets (n_local+1) int32, term_idx (nnz) int32, kept (nnz) float32: torch tensors of this rank (CUDA for nccl);
    local_index: global protein index of each local row.  Returns (offsets (total+1) int32, term_idx, kept) in global
    protein order on `dst`, None elsewhere."""
    return FilteredGatherPlan(local_index, total, offsets.device, dst=dst, group=group).run(offsets, term_idx, kept)


class DenseGatherPlan:
    """Everything about the dense gather that does not depend on the scores, computed ONCE: shard sizes (one tiny
    all_gather), the padded payload / receive buffers, and on `dst` the row index that puts every rank's rows back into
    input order.  `run(block)` is then: copy into the payload, ONE dist.gather, one index_copy_ -- no host sync, no
    Python loop over proteins or ranks' rows (bench.py keeps the plan across steps)."""

    def __init__(self, n_local: int, width: int, local_index, total: int, device, dtype=None, dst: int = 0, group=None,
                 collectives_for_one_rank: bool = False):
        """collectives_for_one_rank: testing aid -- issue the real collectives even in a one-rank group (exercises the RCCL calls
        on a single-GPU box, where two ranks cannot share the device)."""
        import torch
        import torch.distributed as dist
        self.dst, self.group, self.total, self.width = dst, group, int(total), int(width)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.single = self.world == 1 and not (collectives_for_one_rank and dist.is_initialized())
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        dtype = dtype or torch.float32
        self.via_host = self.world > 1 and dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"
        dev = torch.device("cpu") if self.via_host else torch.device(device)   # gloo has no device gather; RCCL gathers in HBM
        self.dev, self.n_local = dev, int(n_local)
        idx = torch.as_tensor(np.asarray(list(local_index), dtype=np.int64), device=dev)
        if self.single:
            self.order = idx
            self.out = torch.empty((self.total, self.width), dtype=dtype, device=dev)
            return
        counts = torch.zeros(self.world, dtype=torch.long, device=dev)
        counts[self.rank] = self.n_local
        dist.all_reduce(counts, group=group)                   # once per plan: agree on shard sizes
        self.counts = [int(c) for c in counts.tolist()]
        self.n_max = max(max(self.counts), 1)
        self.payload = torch.zeros((self.n_max, self.width), dtype=dtype, device=dev)
        pad_idx = torch.full((self.n_max,), -1, dtype=torch.long, device=dev)
        pad_idx[:self.n_local] = idx
        is_dst = self.rank == dst
        ibufs = [torch.empty_like(pad_idx) for _ in range(self.world)] if is_dst else None
        dist.gather(pad_idx, ibufs, dst=dst, group=group)      # once per plan: who owns which input row
        self.recv = self.out = self.src_rows = self.dst_rows = None
        if is_dst:
            self.recv = torch.empty((self.world, self.n_max, self.width), dtype=dtype, device=dev)
            self.recv_list = list(self.recv.unbind(0))
            flat = torch.stack(ibufs).reshape(-1)
            self.src_rows = torch.nonzero(flat >= 0).reshape(-1)
            self.dst_rows = flat[self.src_rows]
            self.out = torch.empty((self.total, self.width), dtype=dtype, device=dev)

    @staticmethod
    def footprint(counts, width: int, itemsize: int = 4):
        """Sizes of the plan's buffers for shard sizes `counts`, without allocating anything (bench.py --dry-plan, tests): elements
        of one rank's padded payload (= one collective's count), of the destination's receive block and of its output; rows are
        indexed with int64 throughout (`index_copy_` / `index_select`), so only the per-collective element count has a limit."""
        n_max, world, total = max(max(counts), 1), len(counts), int(sum(counts))
        return {"payload_elems": n_max * width, "recv_elems": world * n_max * width, "out_elems": total * width,
                "payload_bytes": n_max * width * itemsize, "out_bytes": total * width * itemsize, "row_index_dtype": "int64"}

    def run(self, block):
        """block: (n_local, width) scores of this rank.  Returns the (total, width) tensor in input order on `dst`, None
        elsewhere.  Asynchronous on the device for the nccl backend."""
        import torch.distributed as dist
        if self.single:
            self.out.index_copy_(0, self.order, block.to(self.dev))
            return self.out
        if self.n_local:
            self.payload[:self.n_local].copy_(block, non_blocking=True)
        dist.gather(self.payload, self.recv_list if self.rank == self.dst else None, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        self.out.index_copy_(0, self.dst_rows, self.recv.reshape(-1, self.width).index_select(0, self.src_rows))
        return self.out


class FilteredGatherPlan:
    """The compacted gather (gather_filtered) with its host work hoisted: the first `run` agrees on the sizes (it has to
    read the survivor counts back once); later runs reuse padded buffers with 25 % headroom.  Every payload carries the
    rank's true survivor count in a trailer element, so `dst` learns all sizes from ONE read-back of `world` integers.
    Output as gather_filtered: (offsets (total+1) int32, term_idx, kept) in input order on `dst`.

    Host synchronisations per `run`: sizes_may_change=True -- one on every rank (the size agreement); False -- none on the
    sending ranks, one on `dst` (the trailer read-back that sizes its output)."""

    def __init__(self, local_index, total: int, device, dst: int = 0, group=None, collectives_for_one_rank: bool = False):
        import torch
        import torch.distributed as dist
        self.dst, self.group, self.total = dst, group, int(total)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.single = self.world == 1 and not (collectives_for_one_rank and dist.is_initialized())
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.via_host = self.world > 1 and dist.get_backend(group) == "gloo" and torch.device(device).type == "cuda"
        self.dev = torch.device("cpu") if self.via_host else torch.device(device)
        self.local_index = list(local_index)
        self.z_cap = self.z_cap_last = 0
        self._overflowed = False      # a rank outgrew the plan in an earlier run(sizes_may_change=False): raised by every rank together
        self._pending = None          # (host copy of the all-reduced overflow flag, event) of the previous run
        self.dense = DenseGatherPlan(len(self.local_index), 1, self.local_index, total, self.dev, dtype=torch.int64, dst=dst, group=group,
                                     collectives_for_one_rank=collectives_for_one_rank)

    @staticmethod
    def max_survivors() -> int:
        """The returned offsets are int32 (the output stage's format, mDeepFRI.output): the whole job's survivors must stay below 2^31."""
        return 2 ** 31 - 1

    # -- overflow agreement, one step late: all ranks learn it without a synchronisation inside the step ------------------
    def _post_flag(self, overflow: bool):
        import torch
        import torch.distributed as dist
        flag = torch.full((1,), 1 if overflow else 0, dtype=torch.int32, device=self.dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
        if self.dev.type == "cuda":
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self._pending = (host, ev, flag)
        else:
            self._pending = (flag, None, flag)

    def check(self):
        """Raise -- on EVERY rank -- if some rank's survivors exceeded the planned payload in the previous
        run(sizes_may_change=False); that run's result is incomplete.  Called at the start of each run and by the caller
        after its last one; by then the flag has long arrived, so the wait is not a stall."""
        if self._pending is not None:
            host, ev, _ = self._pending
            self._pending = None
            if ev is not None:
                ev.synchronize()
            if int(host[0]) != 0:
                self._overflowed = True
        if self._overflowed:
            self._overflowed = False
            self.z_cap = 0        # the next run re-plans
            raise RuntimeError(f"rank {self.rank}: a rank's survivors exceeded the planned payload of {self.z_cap_last} entries in the previous "
                               "run; its result is incomplete -- call run(..., sizes_may_change=True)")

    def run(self, offsets, term_idx, kept, sizes_may_change: bool = True):
        """sizes_may_change=True (default, always safe): every call agrees on the padded payload size with one tiny all-reduce and
        one read-back.  False: the caller expects that no rank's survivor count exceeds what the first call planned for (e.g.
        bench.py, which repeats the same workload) -- then the sending ranks never wait for the device.  A rank that outgrows
        the plan truncates its payload, COMPLETES the collectives (nobody hangs), and the overflow is raised on every rank by the
        next run() or check() (the sizes travel with the payload, the flag by an all-reduce read one step late)."""
        import torch
        import torch.distributed as dist
        cnt = (offsets[1:] - offsets[:-1]).to(torch.int64).to(self.dev)
        term_idx, kept = term_idx.to(self.dev), kept.to(self.dev)
        if self.single:
            counts = self.dense.run(cnt.reshape(-1, 1)).reshape(-1)
            goff = torch.zeros(self.total + 1, dtype=torch.int64, device=self.dev)
            torch.cumsum(counts, 0, out=goff[1:])
            return _place_filtered(goff, [(self.dense.order, cnt, term_idx, kept)], self.dev, [int(term_idx.numel())])
        self.check()
        z = int(term_idx.numel())
        zs_host = None
        if self.z_cap == 0 or sizes_may_change:             # a COLLECTIVE decision: every rank takes this branch together
            zs = torch.zeros(self.world, dtype=torch.long, device=self.dev)
            zs[self.rank] = z
            dist.all_reduce(zs, group=self.group)
            zs_host = [int(v) for v in zs.tolist()]
            need = max(zs_host)
            if need > self.z_cap or self.z_cap == 0:
                self.z_cap = max(need * 5 // 4, 1)
                self.t_pay = torch.zeros(self.z_cap + 1, dtype=torch.int32, device=self.dev)     # [z_cap] = the rank's true count
                self.s_pay = torch.zeros(self.z_cap, dtype=torch.float32, device=self.dev)
                if self.rank == self.dst:
                    self.t_recv = [torch.empty_like(self.t_pay) for _ in range(self.world)]
                    self.s_recv = [torch.empty_like(self.s_pay) for _ in range(self.world)]
        self.z_cap_last = self.z_cap
        z_send = min(z, self.z_cap)
        counts = self.dense.run(cnt.reshape(-1, 1))          # per-protein survivor counts in input order (dst only)
        self.t_pay[:z_send].copy_(term_idx[:z_send], non_blocking=True)
        self.s_pay[:z_send].copy_(kept[:z_send], non_blocking=True)
        self.t_pay[self.z_cap] = z
        is_dst = self.rank == self.dst
        dist.gather(self.t_pay, self.t_recv if is_dst else None, dst=self.dst, group=self.group)
        dist.gather(self.s_pay, self.s_recv if is_dst else None, dst=self.dst, group=self.group)
        if zs_host is None:
            self._post_flag(z > self.z_cap)
        if not is_dst:
            return None
        if zs_host is None:                                  # ONE read-back: every rank's true survivor count
            zs_host = [int(v) for v in torch.stack([t[self.z_cap] for t in self.t_recv]).tolist()]
            if max(zs_host) > self.z_cap:                    # truncated payloads cannot be placed; every rank raises in check()
                self.check()
        counts = counts.reshape(-1)
        goff = torch.zeros(self.total + 1, dtype=torch.int64, device=self.dev)
        torch.cumsum(counts, 0, out=goff[1:])
        d = self.dense
        blocks, flat_rows = [], d.dst_rows
        start = 0
        for r in range(self.world):
            n_r = d.counts[r]
            rows = flat_rows[start:start + n_r]
            start += n_r
            blocks.append((rows, counts[rows], self.t_recv[r], self.s_recv[r]))
        return _place_filtered(goff, blocks, self.dev, zs_host)


def _place_filtered(goff, blocks, dev, sizes):
    """Scatter per-rank compacted payloads into global protein order.  blocks: (global rows of the rank's proteins in payload
    order, their counts, term payload, score payload); payloads may carry padding behind the rank's survivors.  `sizes`: the
    survivors of each block as host integers (the caller read them back once), so nothing here waits for the device."""
    import torch
    n = int(sum(sizes))
    if n > FilteredGatherPlan.max_survivors():
        raise OverflowError(f"{n} surviving (term, score) pairs do not fit the int32 offsets of the compacted output; gather in several parts")
    out_t = torch.empty(n, dtype=torch.int32, device=dev)
    out_s = torch.empty(n, dtype=torch.float32, device=dev)
    for (rows, c, t, s), z in zip(blocks, sizes):
        if z == 0:
            continue
        lo = torch.cumsum(c, 0) - c                                   # start of each protein inside the rank's payload
        dest = torch.repeat_interleave(goff[rows] - lo, c, output_size=z) + torch.arange(z, device=dev)
        out_t[dest] = t[:z]
        out_s[dest] = s[:z]
    return goff.to(torch.int32), out_t, out_s


def predict_sharded(engine, seqs, coords, q_alns, t_alns, modes=None, dst: int = 0, group=None, max_rows: int = None):
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
    full = DenseGatherPlan(len(mine), sum(widths), mine, n, engine.device, dst=dst, group=group).run(block)
    if full is None:
        return None
    res, c = {}, 0
    for m, w in zip(modes, widths):
        res[m] = full[:, c:c + w]
        c += w
    return res


def predict_sharded_filtered(engine, seqs, coords, q_alns, t_alns, threshold: float = 0.1, modes=None, dst: int = 0, group=None,
                             max_rows: int = None):
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
        res[m] = FilteredGatherPlan(mine, n, engine.device, dst=dst, group=group).run(off, ti, kept)
    return res if rank == dst else None
