"""Host-buffers-in, host-buffers-out runner for arbitrarily long inputs: the batched counterpart of the reference's two loops
(`Pool.map(build_align_contact_map)`, pipeline.py:476-481, and `_run_prediction_loop`, pipeline.py:292-319) when the
alignments do not fit -- or should not wait -- in one device batch.

A producer thread packs the next batch (PackedProteins.pack: pure host work) while the GPU computes the current one; the
main thread only enqueues: upload, fused forward, asynchronous copy of the scores into pinned host memory.  Results of
batch k are handed out after batch k+1 has been enqueued, so the device never waits for the host between batches.
Everything on the device side is the same HotPathEngine; PyTorch is used for pinned memory, the stream and events only.
"""
from __future__ import annotations

import queue
import threading
from dataclasses import dataclass, field

import numpy as np

from . import _hip
from .batch import HotPathEngine, PackedProteins, DeviceBatch


def _fields(item):
    """(sequence, coords, gapped query, gapped target) of a tuple or of an AlignmentResult-like object
    (reference alignment.py:106-150: query_sequence / coords / gapped_sequence / gapped_target)."""
    if isinstance(item, (tuple, list)):
        return item
    return (item.gapped_sequence.replace("-", ""), item.coords, item.gapped_sequence, item.gapped_target)


class AlignmentStream:
    """engine: a HotPathEngine; batch_size: proteins per device batch (10 000 L=512 proteins keep an MI355X busy for ~0.13 s
    with three GO heads); max_rows: residue rows per fused chunk inside a batch.  sort_by_length (default): a batch goes through the
    path shortest protein first (proteins of like length share chunks: ~10 % on the GCN stage of mixed-length batches, DESIGN.md
    section 5 row 14) and its score rows are put back in input order on the device before they travel -- results and their order do
    not depend on it."""

    def __init__(self, engine: HotPathEngine, batch_size: int = 10000, max_rows: int = None, prefetch: int = 2, sort_by_length: bool = True):
        self.engine = engine
        self.batch_size = int(batch_size)
        self.max_rows = int(max_rows) if max_rows else int(engine.max_rows)   # (default: the engine's chunk size)
        self.prefetch = int(prefetch)
        self.sort_by_length = bool(sort_by_length)

    def _producer(self, items, q):
        try:
            buf, first = [], 0
            for it in items:
                buf.append(_fields(it))
                if len(buf) == self.batch_size:
                    q.put((first, self._pack(buf)))
                    first += len(buf)
                    buf = []
            if buf:
                q.put((first, self._pack(buf)))
            q.put(None)
        except BaseException as e:  # surfaces in the consumer
            q.put(e)

    def _pack(self, rows):
        """-> (packed, order, rows): order[k] = position inside the batch of the k-th packed protein (None: input order)."""
        order = None
        if self.sort_by_length:
            order = sorted(range(len(rows)), key=lambda i: len(rows[i][0]))      # stable: equal lengths keep their input order
            if order == list(range(len(rows))):
                order = None
        use = rows if order is None else [rows[i] for i in order]
        pk = PackedProteins.pack([r[0] for r in use], [r[1] for r in use], [r[2] for r in use], [r[3] for r in use], max_rows=self.max_rows)
        return pk, order, rows

    def run(self, items):
        """Generator over (first_index, {mode: np.ndarray (n, T) float32}) in input order.  Raises what the device flags
        (invalid residue, CSR overflow after one automatic retry with a larger capacity)."""
        import torch
        eng = self.engine
        q = queue.Queue(maxsize=self.prefetch)
        th = threading.Thread(target=self._producer, args=(items, q), daemon=True)
        th.start()
        pending = None   # (first, db, {mode: pinned tensor}, event)
        # the scores travel back on a second, high-priority stream (110 MB per 10 000 proteins and three heads: 4 ms of PCIe): in the
        # compute stream they would hold the next batch up, and streams of equal priority may share a hardware queue
        side = torch.cuda.Stream(eng.device, priority=-1)
        while True:
            got = q.get()
            if isinstance(got, BaseException):
                raise got
            if got is not None:
                first, (pk, order, rows) = got
                with torch.cuda.device(eng.device):
                    main = torch.cuda.current_stream(eng.device)
                    db = eng.upload(pk)        # in the compute stream: 75 MB per 10 000 proteins next to the GEMMs on another stream cost more than they hide
                    out = eng.forward_alignments(db)
                    if order is not None:      # back to input order before the rows travel (a device-side gather: ~30 us per 100 MB)
                        inv = torch.from_numpy(np.argsort(np.asarray(order, dtype=np.int64))).to(eng.device, non_blocking=True)
                        out = {m: t.index_select(0, inv) for m, t in out.items()}
                    done = torch.cuda.Event()
                    done.record(main)
                    host = {m: torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for m, t in out.items()}
                    # the validity flags travel with the scores: reading them later must not queue behind the next batch
                    flags = (torch.empty(db.bad.shape, dtype=db.bad.dtype, pin_memory=True),
                             torch.empty(db.status.shape, dtype=db.status.dtype, pin_memory=True))
                    with torch.cuda.stream(side):
                        side.wait_event(done)
                        for m, t in out.items():
                            t.record_stream(side)
                            host[m].copy_(t, non_blocking=True)
                        flags[0].copy_(db.bad, non_blocking=True)
                        flags[1].copy_(db.status, non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                nxt = (first, db, host, ev, pk, flags, order, rows)
            else:
                nxt = None
            if pending is not None:
                yield self._finish(pending)
            pending = nxt
            if got is None:
                break
        th.join()

    def _finish(self, pending):
        first, db, host, ev, pk, flags, order, rows = pending
        ev.synchronize()
        try:
            self.engine.raise_flags(pk, flags[0].numpy(), flags[1].numpy())
        except _hip.CapacityError:   # rare: a denser batch than the CSR capacity planned for; redo it synchronously
            res = self.engine.run_alignments(pk)
            if order is not None:
                inv = np.argsort(np.asarray(order, dtype=np.int64))
                res = {m: a[inv] for m, a in res.items()}
            return first, res
        except ValueError:
            if order is None:
                raise
            # an invalid residue (or an over-long query): the reference reports the FIRST one in input order -- let the batch in input
            # order say which (error path only)
            self.engine.run_alignments(PackedProteins.pack([r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], [r[3] for r in rows],
                                                           max_rows=self.max_rows))
            raise
        return first, {m: t.numpy() for m, t in host.items()}

    def run_all(self, items) -> dict:
        """Convenience: concatenate every batch -> {mode: np.ndarray (N, T)}."""
        parts = {}
        for _, res in self.run(items):
            for m, a in res.items():
                parts.setdefault(m, []).append(a)
        return {m: np.concatenate(v, axis=0) for m, v in parts.items()}


@dataclass
class QueryBatchResult:
    """One slice of the input after QueryStream: `batch` = the AlignedBatch of the queries that had candidates (`aligned`: their positions
    inside the slice, in the batch's order -- shortest query first unless the stream was made with sort_by_length=False), `kept` = positions inside `batch` whose hit had a structure and `gcn` the filter's arrays for exactly those
    ({mode: (offsets, term_idx, scores)}); `sequence_only` = positions inside the slice without a structure (no candidate, or a hit
    without a trace) and `cnn` the filter's arrays of the sequence-only heads for those (empty without a sequence engine)."""
    first: int
    count: int
    batch: object = None
    aligned: list = field(default_factory=list)
    kept: list = field(default_factory=list)
    gcn: dict = field(default_factory=dict)
    sequence_only: list = field(default_factory=list)
    cnn: dict = field(default_factory=dict)
    gcn_scores: dict = field(default_factory=dict)      # with keep_scores: {mode: float32 (len(kept), T)} / (len(sequence_only), T)
    cnn_scores: dict = field(default_factory=dict)

    def __iter__(self):      # (first, batch, kept, gcn): the structure branch alone
        return iter((self.first, self.batch, self.kept, self.gcn))


_GCN_LUT = np.full(256, 255, dtype=np.uint8)
_GCN_LUT[np.frombuffer(b"-DGULNTKHYWCPVSOIEFXQABZRM", dtype=np.uint8)] = 0      # the residue alphabet of predict.pyx:26 (case-sensitive)


def _first_foreign(seqs, lut):
    """The first character, in list order, that `lut` (256 bytes, 255 = unknown) does not know; None if there is none."""
    for s in seqs:
        raw = np.frombuffer(s.encode("latin-1", "replace"), dtype=np.uint8)
        bad = lut[raw] == 255
        if bad.any():
            return s[int(np.argmax(bad))]
    return None


class QueryStream:
    """The stages either side of the path as ONE stream over arbitrarily many queries: queries + candidate sets -> best hit and
    alignment (GPU aligner, reference alignment.py:223-320) -> C-alpha trace of the hit -> fused contact map + GCN (pipeline.py:476-481,
    292-319) -> the `score >= threshold` filter of results.tsv (pipeline.py:696-705), batch by batch.

    One host thread, one device stream, a software pipeline four batches deep.  Step t ENQUEUES, in this order: the score launch of
    batch t (`mdf_nw_best_hits_begin`), the winners' alignments of batch t-1 (`_align`: needs the scores of t-1, enqueued a step ago),
    upload + contact maps + GCN + filter of batch t-2 (needs its alignments, enqueued a step ago) -- and then collects batch t-3.
    Every wait is for work that sits at least one whole GCN batch further up the stream, so the host never holds the device up, and
    because everything is in ONE stream the aligner's kernels run between two GCN batches, not among their GEMMs (co-resident they cost
    the GEMMs five times the aligner's own time).  Uploads and the collection of results go through a second, high-priority stream: a
    copy must not queue behind the batch in flight (streams of equal priority may share a hardware queue).

    engine: HotPathEngine.  structures: mapping target key -> float32 (Lt, 3) C-alpha trace (`.get`; a hit without one makes its query a
    sequence-only query, as pipeline.py:485 does).  batch_size: queries per device batch.  sequence_engine: a batch.SequenceEngine with
    the CNN heads for the sequence-only queries (optional).  keep_scores: also hand out the full score matrices (`gcn_scores` /
    `cnn_scores`: {mode: float32 (rows, T)}, what `output.write_prediction_matrix` takes) -- 11 KB per protein and three heads more over
    PCIe, on the side stream.  batch_chunks > 0: cut the batches by residue rows instead of by count -- as many queries as fill that many
    chunks of `max_rows` padded rows (a batch of N queries ends in a mostly empty chunk that still costs whole GEMM rounds).
    sort_by_length (default): the queries of one batch go through the path shortest first, as the reference's work list does
    (pipeline.py:529-533 sorts it by length) -- proteins of like length then share chunks, which is what the per-length choice of the
    aggregation kernel is built for (4 000-query batches, three heads, same box: 77.5 k proteins/s sorted against 71.5 k in arrival
    order; profiles/r04_stream_order.txt).  The results do not depend on it (every protein's scores are the same bits in any batch);
    `aligned[k]` is the position inside the slice of batch entry k."""

    def __init__(self, engine: HotPathEngine, structures, batch_size: int = 4000, max_rows: int = None, scoring_matrix="VTML80",
                 gap_open: int = 10, gap_extend: int = 1, threshold: float = 0.1, capacity_per_protein: int = 64, sequence_engine=None,
                 keep_scores: bool = False, batch_chunks: int = 0, sort_by_length: bool = True):
        import torch
        from .alignment import AlignerWorkspace
        self.batch_chunks, self.sort_by_length = int(batch_chunks), bool(sort_by_length)
        self.engine, self.structures, self.sequence_engine = engine, structures, sequence_engine
        self.batch_size, self.max_rows = int(batch_size), int(max_rows) if max_rows else int(engine.max_rows)
        self.scoring_matrix, self.gap_open, self.gap_extend = scoring_matrix, int(gap_open), int(gap_extend)
        self.threshold, self.capacity_per_protein, self.keep_scores = float(threshold), int(capacity_per_protein), bool(keep_scores)
        self.main, self.side = torch.cuda.Stream(engine.device), torch.cuda.Stream(engine.device, priority=-1)
        self.ring = [AlignerWorkspace(engine.device.index or 0, stream=self.main.cuda_stream) for _ in range(3)]

    def run(self, query_ids, query_sequences, target_sequences):
        """Generator over QueryBatchResult (it also unpacks as (first_index, AlignedBatch, kept, {mode: (offsets, term_idx, scores)})) in
        input order: `kept` = positions inside the batch that had a structure; the arrays (numpy) are the filter's output for those
        proteins, what mDeepFRI.output.results_text takes.  A query with an empty candidate set, or whose best hit has no structure,
        is a sequence-only query: with a `sequence_engine` it runs through the CNN heads in the same step (reference
        pipeline.py:600-648: `unaligned_queries`), without one it is only listed in `sequence_only`."""
        import torch
        from .alignment import align_queries_begin
        query_ids, query_sequences, target_sequences = list(query_ids), list(query_sequences), list(target_sequences)
        if self.batch_chunks > 0 and query_ids:      # slices of ~batch_chunks full chunks: greedy over the padded rows, as the planner fills chunks
            g = int(self.engine.L.mdf_group_rows())      # MDF_GROUP_ROWS: the planner pads every protein to a multiple of it
            rows = (np.fromiter(map(len, query_sequences), dtype=np.int64, count=len(query_sequences)) + g - 1) // g * g
            starts, chunk_rows, chunks = [0], 0, 1
            for i, r in enumerate(rows.tolist()):
                if chunk_rows and chunk_rows + r > self.max_rows:
                    chunks, chunk_rows = chunks + 1, 0
                    if chunks > self.batch_chunks:
                        starts.append(i)
                        chunks = 1
                chunk_rows += r
            ends = starts[1:] + [len(query_ids)]
        else:
            starts = list(range(0, len(query_ids), self.batch_size))
            ends = [min(a + self.batch_size, len(query_ids)) for a in starts]
        nb = len(starts)
        aligning, running, slices = {}, {}, {}
        try:
            with torch.cuda.device(self.engine.device):
                for t in range(nb + 3):
                    if t < nb:
                        a, b = starts[t], ends[t]
                        pos = [i for i in range(a, b) if len(target_sequences[i]) > 0]
                        if self.sort_by_length:      # batch entry k is query pos[k]: `aligned` hands the order out
                            pos.sort(key=lambda i: len(query_sequences[i]))
                        slices[t] = (a, b, pos)
                        if len(pos) == b - a and not self.sort_by_length:
                            args = (query_ids[a:b], query_sequences[a:b], target_sequences[a:b])
                        else:
                            args = ([query_ids[i] for i in pos], [query_sequences[i] for i in pos], [target_sequences[i] for i in pos])
                        aligning[t] = align_queries_begin(*args, self.gap_open, self.gap_extend, self.scoring_matrix, workspace=self.ring[t % 3]) if pos else None
                    if 0 <= t - 1 < nb and aligning[t - 1] is not None:
                        try:
                            aligning[t - 1].launch_alignments()
                        except ValueError as e:
                            self._aligner_error_in_input_order(e, slices[t - 1][2], query_sequences, target_sequences)
                            raise
                    if 0 <= t - 2 < nb:
                        a, b, pos = slices.pop(t - 2)
                        pend = aligning.pop(t - 2)
                        try:
                            batch = pend.result() if pend is not None else None
                        except ValueError as e:
                            self._aligner_error_in_input_order(e, pos, query_sequences, target_sequences)
                            raise
                        running[t - 2] = self._enqueue(a, b, pos, batch, query_ids, query_sequences)
                    if 0 <= t - 3 < nb:
                        yield self._finish(running.pop(t - 3))
        finally:
            for p in aligning.values():      # an exception (or an abandoned generator) leaves batches in flight: drop them
                if p is not None:
                    p.abandon()
            if running:                      # ... and let the enqueued GCN batches finish before their buffers go back to the allocator
                self.main.synchronize()
                running.clear()

    def _aligner_error_in_input_order(self, err, pos, query_sequences, target_sequences):
        """A batch goes through the aligner shortest query first, so the character the library names is the first foreign one in THAT
        order.  AlignmentStream and the unsorted stream name the first one in input order: say the same here (error path only)."""
        if not self.sort_by_length or "is not in the scoring matrix alphabet" not in str(err):
            return
        from .alignment import _matrix
        lut = _matrix(self.scoring_matrix)._lut_nocase
        ordered = sorted(pos)
        c = _first_foreign([query_sequences[i] for i in ordered], lut)
        if c is None:      # (the staged list holds the queries first, then the candidates in first-seen order)
            c = _first_foreign(list(dict.fromkeys(t for i in ordered for t in target_sequences[i].values())), lut)
        if c is not None:
            raise ValueError(f"character {c!r} is not in the scoring matrix alphabet") from None

    def _launch(self, pk, forward):
        """Upload (side stream), `forward(db)` + filter (main stream), the small results on their way back.  -> the part's state."""
        import torch
        from .output import filter_scores_async
        eng, main, side = self.engine, self.main, self.side
        with torch.cuda.stream(side):      # the upload does not queue behind the batch in flight
            db = DeviceBatch(pk, eng.device)
        main.wait_stream(side)
        with torch.cuda.stream(main):
            out = forward(db)
            filt = {m: filter_scores_async(t, self.threshold, db.B * self.capacity_per_protein) for m, t in out.items()}
        # NO copy back is enqueued here: a device-to-host copy waiting in the compute stream behind a whole batch also holds up the
        # copy engine's queue, and with it every later copy of the side stream (measured: the collection of batch k waited for batch k+1)
        return db, pk, out, filt

    def _enqueue(self, a, b, pos, batch, query_ids, query_sequences):
        import torch
        res = QueryBatchResult(first=a, count=b - a, batch=batch, aligned=[i - a for i in pos])
        structured = set()
        gcn = cnn = None
        if batch is not None:
            coords = [self.structures.get(k) for k in batch.target_keys]
            if any(c is not None for c in coords):
                pk, res.kept = PackedProteins.from_aligned_batch(batch, coords, max_rows=self.max_rows)
                structured = {res.aligned[k] for k in res.kept}
                gcn = self._launch(pk, self.engine.forward_alignments)
        res.sequence_only = [i for i in range(b - a) if i not in structured]
        if self.sequence_engine is not None and res.sequence_only:
            pk = PackedProteins.pack([query_sequences[a + i] for i in res.sequence_only], max_rows=self.sequence_engine.max_rows)
            cnn = self._launch(pk, self.sequence_engine.forward)
        ev = None
        if gcn is not None or cnn is not None:
            ev = torch.cuda.Event()
            ev.record(self.main)
        return res, gcn, cnn, ev

    def _collect(self, part, redo):
        """Results of one part (after its event): {mode: (offsets, term_idx, scores)} as numpy, through the side stream."""
        import torch
        from .batch import first_invalid_residue
        from .output import filter_scores
        db, pk, out, filt = part
        eng, side = self.engine, self.side
        with torch.cuda.stream(side):      # the part's event has been waited for: its small results first (flags, offsets, filter status)
            flags = (torch.empty(db.bad.shape, dtype=db.bad.dtype, pin_memory=True), torch.empty(db.status.shape, dtype=db.status.dtype, pin_memory=True))
            flags[0].copy_(db.bad, non_blocking=True)
            flags[1].copy_(db.status, non_blocking=True)
            small = {m: (torch.empty(f[0].shape, dtype=f[0].dtype, pin_memory=True), torch.empty(4, dtype=torch.int32, pin_memory=True))
                     for m, f in filt.items()}
            for m, f in filt.items():
                f[0].record_stream(side)
                f[3].record_stream(side)
                small[m][0].copy_(f[0], non_blocking=True)
                small[m][1].copy_(f[3], non_blocking=True)
            side.synchronize()
        try:
            if redo is None:      # sequence-only models flag invalid residues only
                first_invalid_residue(pk, flags[0].numpy())
            else:
                eng.raise_flags(pk, flags[0].numpy(), flags[1].numpy())
        except _hip.CapacityError:   # rare: a denser batch than the CSR capacity planned for; redo it synchronously
            # the engine's workspaces are ONE set: the batches already enqueued behind this one must have left them before the batch is
            # run again (with a larger capacity), and the re-run goes into the same stream as everything else
            self.main.synchronize()
            with torch.cuda.stream(self.main):
                res, redone = {}, redo(pk)
                for m, arr in redone.items():
                    o, ti, sc = filter_scores(torch.from_numpy(arr).to(eng.device), self.threshold, self.capacity_per_protein)
                    res[m] = (o.cpu().numpy(), ti.cpu().numpy(), sc.cpu().numpy())
            return res, (redone if self.keep_scores else {})
        res = {}
        with torch.cuda.stream(side):      # not behind the next batch, which already occupies the main stream
            for m, (offsets, term_idx, sc, _) in filt.items():
                off_h, st_h = small[m][0].numpy(), small[m][1].numpy()
                if st_h[0] != 0:        # more survivors than capacity_per_protein allowed for: filter this head again, sized exactly
                    out[m].record_stream(side)
                    o, ti, s2 = filter_scores(out[m], self.threshold, int(st_h[1]) // max(db.B, 1) + 1)
                    res[m] = (o.cpu().numpy(), ti.cpu().numpy(), s2.cpu().numpy())
                    continue
                n = int(off_h[-1])
                term_idx.record_stream(side)
                sc.record_stream(side)
                ti_h, sc_h = torch.empty(n, dtype=torch.int32, pin_memory=True), torch.empty(n, dtype=torch.float32, pin_memory=True)
                ti_h.copy_(term_idx[:n], non_blocking=True)
                sc_h.copy_(sc[:n], non_blocking=True)
                res[m] = (off_h, ti_h.numpy(), sc_h.numpy())
            full = {}
            if self.keep_scores:
                for m, t in out.items():
                    t.record_stream(side)
                    full[m] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    full[m].copy_(t, non_blocking=True)
            side.synchronize()
        return res, {m: t.numpy() for m, t in full.items()}

    def _finish(self, pending):
        res, gcn, cnn, ev = pending
        if ev is not None:
            ev.synchronize()
        if gcn is not None:
            try:
                res.gcn, res.gcn_scores = self._collect(gcn, self.engine.run_alignments)
            except ValueError as e:
                if self.sort_by_length and str(e).startswith("Invalid character in sequence"):
                    # the packed batch is in sorted order; name the first invalid residue of the slice in INPUT order, as AlignmentStream does
                    seqs = gcn[1].seqs
                    by_input = sorted(range(len(seqs)), key=lambda k: res.aligned[res.kept[k]])
                    c = _first_foreign([seqs[k] for k in by_input], _GCN_LUT)
                    if c is not None:
                        raise ValueError(f"Invalid character in sequence: {c}") from None
                raise
        if cnn is not None:
            res.cnn, res.cnn_scores = self._collect(cnn, None)
        return res
