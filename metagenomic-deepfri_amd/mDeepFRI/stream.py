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

import numpy as np

from . import _hip
from .batch import HotPathEngine, PackedProteins


def _fields(item):
    """(sequence, coords, gapped query, gapped target) of a tuple or of an AlignmentResult-like object
    (reference alignment.py:106-150: query_sequence / coords / gapped_sequence / gapped_target)."""
    if isinstance(item, (tuple, list)):
        return item
    return (item.gapped_sequence.replace("-", ""), item.coords, item.gapped_sequence, item.gapped_target)


class AlignmentStream:
    """engine: a HotPathEngine; batch_size: proteins per device batch (10 000 L=512 proteins keep an MI355X busy for ~0.18 s
    with three GO heads); max_rows: residue rows per fused chunk inside a batch."""

    def __init__(self, engine: HotPathEngine, batch_size: int = 10000, max_rows: int = 65536, prefetch: int = 2):
        self.engine = engine
        self.batch_size = int(batch_size)
        self.max_rows = int(max_rows)
        self.prefetch = int(prefetch)

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
        return PackedProteins.pack([r[0] for r in rows], [r[1] for r in rows], [r[2] for r in rows], [r[3] for r in rows],
                                   max_rows=self.max_rows)

    def run(self, items):
        """Generator over (first_index, {mode: np.ndarray (n, T) float32}) in input order.  Raises what the device flags
        (invalid residue, CSR overflow after one automatic retry with a larger capacity)."""
        import torch
        eng = self.engine
        q = queue.Queue(maxsize=self.prefetch)
        th = threading.Thread(target=self._producer, args=(items, q), daemon=True)
        th.start()
        pending = None   # (first, db, {mode: pinned tensor}, event)
        while True:
            got = q.get()
            if isinstance(got, BaseException):
                raise got
            if got is not None:
                first, pk = got
                with torch.cuda.device(eng.device):
                    db = eng.upload(pk)
                    out = eng.forward_alignments(db)
                    host = {m: torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for m, t in out.items()}
                    for m, t in out.items():
                        host[m].copy_(t, non_blocking=True)
                    # the validity flags travel with the scores: reading them later must not queue behind the next batch
                    flags = (torch.empty(db.bad.shape, dtype=db.bad.dtype, pin_memory=True),
                             torch.empty(db.status.shape, dtype=db.status.dtype, pin_memory=True))
                    flags[0].copy_(db.bad, non_blocking=True)
                    flags[1].copy_(db.status, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(eng.device))
                nxt = (first, db, host, ev, pk, flags)
            else:
                nxt = None
            if pending is not None:
                yield self._finish(pending)
            pending = nxt
            if got is None:
                break
        th.join()

    def _finish(self, pending):
        first, db, host, ev, pk, flags = pending
        ev.synchronize()
        try:
            self.engine.raise_flags(pk, flags[0].numpy(), flags[1].numpy())
        except _hip.CapacityError:   # rare: a denser batch than the CSR capacity planned for; redo it synchronously
            return first, self.engine.run_alignments(pk)
        return first, {m: t.numpy() for m, t in host.items()}

    def run_all(self, items) -> dict:
        """Convenience: concatenate every batch -> {mode: np.ndarray (N, T)}."""
        parts = {}
        for _, res in self.run(items):
            for m, a in res.items():
                parts.setdefault(m, []).append(a)
        return {m: np.concatenate(v, axis=0) for m, v in parts.items()}
