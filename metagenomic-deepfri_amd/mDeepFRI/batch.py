"""Batched counterparts of the reference's per-protein loops, built on the device entry points of mdfri.h.

reference                                               here
---------                                               ----
pipeline.py:476-481  Pool.map(build_align_contact_map)  build_align_contact_maps()      (dense int32 maps out)
pipeline.py:292-319  _run_prediction_loop               HotPathEngine.forward_*() + prediction_rows()

`HotPathEngine.forward_alignments` is the fused path: C-alpha coordinates + gapped alignments + sequences go in,
GO scores come out; the (L,L) maps never exist, the adjacency goes from the contact kernel to the GraphConv kernels
as CSR in HBM.  PyTorch is used only to own device memory and the stream.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import numpy as np

from . import _hip


def _torch():
    import torch
    return torch


# ---------------------------------------------------------------------------------------------------------------------
# host-side packing
# ---------------------------------------------------------------------------------------------------------------------
def _is_f32(a) -> bool:
    return getattr(a, "dtype", None) == np.float32


def _offsets(lengths) -> np.ndarray:
    off = np.zeros(len(lengths) + 1, dtype=np.int64)
    np.cumsum(np.asarray(lengths, dtype=np.int64), out=off[1:])
    if off[-1] >= 2**31 - 1:
        raise ValueError("batch too large for int32 offsets; split it")
    return off.astype(np.int32)


@dataclass
class Chunk:
    p0: int
    p1: int
    rows: int              # R of this chunk (multiple of 128)
    row_off_pos: int       # start of this chunk's row_off (p1-p0+1 entries) in PackedProteins.chunk_row_off
    segment: int = 0       # pooling segment this chunk belongs to
    group_base: int = 0    # first pooling group (MDF_GROUP_ROWS = 16 rows) of this chunk inside its segment's partial-sum array


@dataclass
class Segment:
    """Consecutive chunks whose per-group partial sums share one array and are pooled by one launch per GO head."""
    p0: int
    p1: int
    groups: int            # pooling groups (16 rows each) in the segment
    grp_off_pos: int       # start of the segment's grp_off (p1-p0+1 entries) in PackedProteins.grp_off


@dataclass
class PackedProteins:
    """Proteins packed into flat arrays (host).  Coordinates / alignments are optional (dense-map path)."""
    seqs: list
    Lq: np.ndarray
    seq_bytes: np.ndarray
    seq_off: np.ndarray
    coords: np.ndarray | None = None
    coord_off: np.ndarray | None = None
    q_aln: np.ndarray | None = None
    t_aln: np.ndarray | None = None
    aln_off: np.ndarray | None = None
    chunks: list = field(default_factory=list)
    chunk_row_off: np.ndarray | None = None
    segments: list = field(default_factory=list)
    grp_off: np.ndarray | None = None
    order: np.ndarray | None = None    # plan position -> index in this batch (mdf_plan_order); None = the plan keeps the input order

    @property
    def B(self) -> int:
        return len(self.seqs)

    @classmethod
    def pack(cls, seqs, coords=None, q_alns=None, t_alns=None, max_rows: int = None, max_segment_groups: int = 1 << 20, keep_order: bool = False):
        """keep_order=False (default): the library's plan VISITS the proteins shortest first (the reference sorts its work list by length,
        pipeline.py:529-533; ~10 % on the GCN stage of a mixed-length batch that arrives unsorted); the packed arrays, the scores and every
        report stay in the order given here -- only `chunks` / `segments` / `order` speak of plan positions.  keep_order=True: visited as given.

        Host-side packing, vectorised: one join + one encode per column (a non-ASCII letter fails there, as `str.encode("ascii")`
        per sequence did), lengths by `map(len, ...)`, the "gapped query spells its sequence" check by one reduceat over the packed
        bytes -- no per-protein Python work besides `len` (the producer thread of mDeepFRI.stream packs batches while the GPU
        computes: at 22 us per protein the old loop, not the GPU, bounded the host-to-host rate)."""
        seqs = list(seqs)
        if not seqs:
            raise ValueError("empty batch")
        Lq = np.fromiter(map(len, seqs), dtype=np.int64, count=len(seqs))
        if (Lq <= 0).any():
            raise ValueError("empty sequence in batch")
        seq_bytes = np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).copy()
        pk = cls(seqs=seqs, Lq=Lq.astype(np.int32), seq_bytes=seq_bytes, seq_off=_offsets(Lq))
        if coords is not None:
            if q_alns is None or t_alns is None or not (len(coords) == len(q_alns) == len(t_alns) == len(seqs)):
                raise ValueError("coords, q_alns and t_alns must all be given, one per sequence")
            cs = [c if type(c) is np.ndarray else np.asarray(c) for c in coords]
            for c in cs:
                if c.dtype != np.float32 or c.ndim != 2 or c.shape[1] != 3:
                    raise ValueError("coordinates must be float32 (Lt,3)")
            pk.coords = np.ascontiguousarray(np.concatenate(cs, axis=0)) if cs else np.zeros((0, 3), np.float32)
            if pk.coords.shape[0] == 0:
                pk.coords = np.zeros((1, 3), np.float32)
            pk.coord_off = _offsets([c.shape[0] for c in cs])
            La = np.fromiter(map(len, q_alns), dtype=np.int64, count=len(seqs))
            Lt_aln = np.fromiter(map(len, t_alns), dtype=np.int64, count=len(seqs))
            if not np.array_equal(La, Lt_aln):
                i = int(np.argmax(La != Lt_aln))
                raise ValueError(f"protein {i}: gapped query and target differ in length")
            pk.q_aln = np.frombuffer("".join(q_alns).encode("ascii"), dtype=np.uint8).copy()
            pk.t_aln = np.frombuffer("".join(t_alns).encode("ascii"), dtype=np.uint8).copy()
            pk.aln_off = _offsets(La)
            gaps = np.flatnonzero(pk.q_aln == 45)                        # gap columns are few: count them per protein by bisection
            nongap = La - (np.searchsorted(gaps, pk.aln_off[1:]) - np.searchsorted(gaps, pk.aln_off[:-1]))
            if not np.array_equal(nongap, Lq):
                i = int(np.argmax(nongap != Lq))
                raise ValueError(f"protein {i}: gapped query does not spell a sequence of length {int(Lq[i])}")
        pk._plan(max_rows, max_segment_groups, keep_order)
        return pk

    @classmethod
    def from_alignments(cls, alignments, max_rows: int = None, max_segment_groups: int = 1 << 20):
        """Pack objects carrying the AlignmentResult attributes the reference's path reads (query_sequence, coords,
        gapped_sequence, gapped_target; reference alignment.py:106-150).  Entries without coordinates are skipped, as
        pipeline.py:485 filters them; returns (packed, kept_indices)."""
        keep = [i for i, a in enumerate(alignments) if a.coords is not None]
        al = [alignments[i] for i in keep]
        pk = cls.pack([a.gapped_sequence.replace("-", "") for a in al], [a.coords for a in al], [a.gapped_sequence for a in al],
                      [a.gapped_target for a in al], max_rows=max_rows, max_segment_groups=max_segment_groups)
        return pk, keep

    @classmethod
    def from_aligned_batch(cls, batch, coords, max_rows: int = None, max_segment_groups: int = 1 << 20):
        """Pack the struct-of-arrays output of mDeepFRI.alignment.align_queries_arrays (the GPU aligner) together with the
        targets' C-alpha coordinates -- no AlignmentResult objects, no per-protein string handling: the gapped strings are
        taken as the flat byte arrays the aligner produced.  coords: one float32 (Lt, 3) array per query (its best target's
        trace), None where the structure is missing; returns (packed, kept_indices) like from_alignments."""
        keep = [i for i, c in enumerate(coords) if c is not None]
        if not keep:
            raise ValueError("empty batch")
        off = np.asarray(batch.aln_off, dtype=np.int64)
        ln_all = off[1:] - off[:-1]
        if len(keep) == len(coords):        # the usual case: every hit has a structure, the aligner's arrays are taken as they are
            ln, seqs, cs = ln_all, list(batch.query_sequences), list(coords)
            q_aln, t_aln = np.ascontiguousarray(batch.q_aln[:int(off[-1])]), np.ascontiguousarray(batch.t_aln[:int(off[-1])])
        else:
            mask = np.zeros(len(coords), dtype=bool)
            mask[keep] = True
            ln = ln_all[mask]
            cols = np.repeat(mask, ln_all)
            q_aln, t_aln = batch.q_aln[:int(off[-1])][cols], batch.t_aln[:int(off[-1])][cols]
            seqs, cs = [batch.query_sequences[i] for i in keep], [coords[i] for i in keep]
        Lq = np.fromiter(map(len, seqs), dtype=np.int32, count=len(seqs))
        if (Lq <= 0).any():
            raise ValueError("empty sequence in batch")
        starts = np.concatenate(([0], np.cumsum(ln)))
        nongap = np.add.reduceat(q_aln != 45, starts[:-1], dtype=np.int64) if len(ln) else np.zeros(0, np.int64)
        if not np.array_equal(nongap, Lq):
            raise ValueError("gapped queries do not spell the query sequences")
        pk = cls(seqs=seqs, Lq=Lq, seq_bytes=np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).copy(), seq_off=_offsets(Lq))
        try:                                  # one concatenation checks every trace at once; a wrong one is looked for only on failure
            xyz = np.concatenate(cs, axis=0)
            ok = xyz.dtype == np.float32 and xyz.ndim == 2 and xyz.shape[1] == 3 and all(map(_is_f32, cs))
        except (ValueError, TypeError):
            ok = False
        if not ok:
            raise ValueError("coordinates must be float32 (Lt,3)")
        pk.coords = np.ascontiguousarray(xyz)
        pk.coord_off = _offsets(np.fromiter(map(len, cs), dtype=np.int64, count=len(cs)))
        pk.q_aln, pk.t_aln, pk.aln_off = q_aln, t_aln, _offsets(ln)
        pk._plan(max_rows, max_segment_groups)
        return pk, keep

    def _plan(self, max_rows: int, max_segment_groups: int = 1 << 20, keep_order: bool = False):
        """Chunks and pooling segments from the library's planner (mdf_plan_create, csrc/engine.hip): the Python objects below
        are a read-only view of its tables; the handle itself is what the engine entry points take."""
        max_rows = int(max_rows) if max_rows else _hip.default_chunk_rows()   # (None / 0: the library's default, MDF_DEFAULT_CHUNK_ROWS)
        import weakref
        L = _hip.lib()
        lq = np.ascontiguousarray(self.Lq, dtype=np.int32)
        h = ctypes.c_void_p()
        _hip.check(L.mdf_plan_create_ex(_hip.ptr(lq), len(lq), int(max_rows), int(max_segment_groups), _hip.MDF_PLAN_KEEP_ORDER if keep_order else 0,
                                        ctypes.byref(h)))
        self.plan = h
        n_order = _hip.c_int64(0)
        optr = L.mdf_plan_order(h, n_order)
        self.order = np.ctypeslib.as_array(optr, shape=(n_order.value,)).copy() if n_order.value else None
        weakref.finalize(self, L.mdf_plan_free, h)
        nc, ns = L.mdf_plan_num_chunks(h), L.mdf_plan_num_segments(h)
        ct, stab = np.empty((nc, 6), dtype=np.int64), np.empty((ns, 4), dtype=np.int64)
        _hip.check(L.mdf_plan_chunks(h, _hip.ptr(ct)))
        _hip.check(L.mdf_plan_segments(h, _hip.ptr(stab)))
        self.chunks = [Chunk(int(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4]), int(r[5])) for r in ct]
        self.segments = [Segment(int(r[0]), int(r[1]), int(r[2]), int(r[3])) for r in stab]
        n = _hip.c_int64(0)
        ptr = L.mdf_plan_chunk_row_off(h, n)
        self.chunk_row_off = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()
        ptr = L.mdf_plan_grp_off(h, n)
        self.grp_off = np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()

    @property
    def max_chunk_rows(self) -> int:
        return max(c.rows for c in self.chunks)


# ---------------------------------------------------------------------------------------------------------------------
# device engine
# ---------------------------------------------------------------------------------------------------------------------
def _p(t, elem_offset: int = 0):
    return ctypes.c_void_p(t.data_ptr() + elem_offset * t.element_size())


# Rows per chunk of the dense-map path (HotPathEngine.forward_dense): its upload slots grow with chunk rows x L (see there); 65 536 keeps
# the double-buffered upload busy (bench.py's `dense` leg) at a quarter of the pinned memory of the fused path's default chunk.
DENSE_CHUNK_ROWS = 65536


class DeviceBatch:
    """PackedProteins uploaded to one GPU (torch tensors own the memory; `desc` is the mdf_batch_dev the engine takes)."""

    def __init__(self, packed: PackedProteins, device):
        torch = _torch()
        self.packed = packed
        self.device = device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=False)  # noqa: E731
        self.Lq = up(packed.Lq)
        self.seq_bytes = up(packed.seq_bytes)
        self.seq_off = up(packed.seq_off)
        self.chunk_row_off = up(packed.chunk_row_off)
        self.grp_off = up(packed.grp_off)
        self.coords = self.coord_off = self.q_aln = self.t_aln = self.aln_off = None
        if packed.coords is not None:
            self.coords, self.coord_off = up(packed.coords), up(packed.coord_off)
            self.q_aln, self.t_aln, self.aln_off = up(packed.q_aln), up(packed.t_aln), up(packed.aln_off)
        n = len(packed.chunks)
        self.status = torch.zeros((n, 4), dtype=torch.int32, device=device)
        self.bad = torch.full((n,), -1, dtype=torch.int64, device=device)   # per chunk: (protein << 32 | position) of the first invalid byte
        a = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        self.desc = _hip.BatchDev(packed.B, a(self.seq_bytes), a(self.seq_off), a(self.Lq), a(self.coords), a(self.coord_off), a(self.q_aln),
                                  a(self.t_aln), a(self.aln_off), a(self.status), a(self.bad))

    @property
    def B(self):
        return self.packed.B


_RESIDUES = np.zeros(256, dtype=bool)
_RESIDUES[np.frombuffer(b"-DGULNTKHYWCPVSOIEFXQABZRM", dtype=np.uint8)] = True      # the alphabet of predict.pyx:26


def first_invalid_residue(packed: PackedProteins, bad) -> None:
    """Raise the reference's ValueError for the first invalid residue of the batch IN THE ORDER IT WAS PACKED, if mdf_seq_encode_dev flagged
    any (the flags are per chunk of the plan, which may visit the proteins shortest first: the sequences themselves say which one comes
    first -- error path only)."""
    if not (np.asarray(bad) != -1).any():
        return
    ok = _RESIDUES[packed.seq_bytes]
    if ok.all():      # (flags without an invalid byte: cannot happen; keep the flag's own answer)
        for ci, ch in enumerate(packed.chunks):
            key = int(bad[ci])
            if key != -1:
                p = ch.p0 + (key >> 32)
                p = int(packed.order[p]) if packed.order is not None else p
                raise ValueError(f"Invalid character in sequence: {packed.seqs[p][key & 0xffffffff]}")
    raise ValueError(f"Invalid character in sequence: {chr(int(packed.seq_bytes[int(np.argmin(ok))]))}")


class HotPathEngine:
    """contact map + GCN forward for batches of proteins on one GPU, for one or several GO heads
    (`predictors`: {mode: mDeepFRI.predict.Predictor}; all heads share the contact-map stage).

    A thin caller of the library's batch engine (include/mdfri.h `mdf_engine_*`, csrc/engine.hip): the planner, the per-chunk
    launch sequence, the language-model grouping, the GO heads and the hipGraph replay of short batches all run in C++; this
    class owns the torch tensors (inputs, flags, outputs) and turns return codes into the exceptions of the per-call API."""

    def __init__(self, predictors: dict, device: int = 0, max_rows: int = None, nnz_per_row: int = 40,
                 threshold: float = 6.0, generated_contacts: int = 2, lm_batch: int = 0, lm_workspace_gib: float = 48.0,
                 graph_max_chunks: int = 0, pipeline_contact: int = 0):
        import weakref
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("HotPathEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.L = _hip.lib()
        self.predictors = dict(predictors)
        self.modes = list(self.predictors)
        self.device = torch.device(f"cuda:{device}")
        self.max_rows = int(max_rows) if max_rows else _hip.default_chunk_rows()
        self.nnz_per_row = int(nnz_per_row)
        self.threshold = float(threshold)
        self.generated_contacts = int(generated_contacts)
        for m, p in self.predictors.items():
            if p.session.kind != "gcn":
                raise ValueError(f"predictor {m!r} is not a GCN model (sequence-only models run on SequenceEngine)")
        # distinct language models, in the engine's order (inspection: lm_features(which=...))
        self.lms = []
        for p in self.predictors.values():
            lm = getattr(p.session, "lm", None)
            if lm is not None and all(lm is not x for x in self.lms):
                self.lms.append(lm)
        cfg = _hip.EngineConfig(self.max_rows, self.nnz_per_row, self.threshold, self.generated_contacts, 0, int(lm_batch), float(lm_workspace_gib),
                                int(graph_max_chunks), int(pipeline_contact))
        handles = (ctypes.c_void_p * len(self.modes))(*[self.predictors[m].session.handle for m in self.modes])
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(self.L.mdf_engine_create(handles, len(self.modes), int(device), ctypes.byref(cfg), ctypes.byref(h)))
        self.handle = h
        weakref.finalize(self, self.L.mdf_engine_free, h)

    def upload(self, packed: PackedProteins) -> DeviceBatch:
        return DeviceBatch(packed, self.device)

    def _stream(self):
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _outputs(self, db: DeviceBatch, want_logits: bool):
        torch = _torch()
        scores = {m: torch.empty((db.B, p.n_terms), dtype=torch.float32, device=self.device) for m, p in self.predictors.items()}
        logits = {m: torch.empty((db.B, 2 * p.n_terms), dtype=torch.float32, device=self.device) for m, p in self.predictors.items()} if want_logits else None
        sp = (ctypes.c_void_p * len(self.modes))(*[scores[m].data_ptr() for m in self.modes])
        lp = (ctypes.c_void_p * len(self.modes))(*[logits[m].data_ptr() for m in self.modes]) if want_logits else None
        return scores, logits, sp, lp

    def forward_alignments(self, db: DeviceBatch, want_logits: bool = False, out=None):
        """Fused path: coords + alignments + sequences -> {mode: (B,T) float32 scores on the device}.  Asynchronous
        on the current stream; call `check(db)` (one sync) before trusting the result.  `out`: the tuple a previous call with
        the same `db` returned through `outputs_for(db)` -- re-using the output tensors lets the engine replay a short batch
        as one hipGraph."""
        if db.coords is None:
            raise ValueError("batch was packed without coordinates/alignments")
        torch = _torch()
        with torch.cuda.device(self.device):
            scores, logits, sp, lp = out if out is not None else self._outputs(db, want_logits)
            _hip.check(self.L.mdf_engine_forward_alignments(self.handle, db.packed.plan, ctypes.byref(db.desc), sp, lp, self._stream()))
        return (scores, logits) if logits is not None else scores

    def outputs_for(self, db: DeviceBatch, want_logits: bool = False):
        """Pre-allocated outputs for repeated forward_alignments(db, out=...) calls (steady-state serving of one batch shape)."""
        return self._outputs(db, want_logits)

    def forward_dense(self, db: DeviceBatch, cmaps, want_logits: bool = False):
        """Reference-format path: one dense (L,L) contact map per protein (what build_align_contact_map returns,
        int32) -> scores.  Maps are uploaded chunk by chunk (inside the library); synchronises before returning.

        The library stages a chunk's maps in two pinned host slots and two device slots of (sum of L^2 over the chunk's proteins) x 4 B
        each -- about chunk rows x L x 4 B: 128 MiB per slot at 65 536 rows of L = 512, four times that at the fused path's default
        chunk (MDF_DEFAULT_CHUNK_ROWS) -- and the first chunk's host copy overlaps nothing.  A batch planned with larger chunks
        is therefore re-planned here with DENSE_CHUNK_ROWS rows per chunk (a second plan + upload of the small per-protein arrays; the
        scores do not depend on the plan: tests/test_gpu_engine.py); plan with `max_rows <= DENSE_CHUNK_ROWS` to avoid the copy."""
        torch = _torch()
        if len(cmaps) != db.B:
            raise ValueError("one contact map per protein expected")
        if db.packed.max_chunk_rows > DENSE_CHUNK_ROWS:
            db = self._dense_view(db)
        flat = []
        for p, A in enumerate(cmaps):
            A = np.asarray(A)
            Lp = int(db.packed.Lq[p])
            if A.shape != (Lp, Lp):
                raise ValueError(f"protein {p}: cmap shape {A.shape} != ({Lp},{Lp})")
            flat.append(np.ascontiguousarray(A, dtype=np.float32 if A.dtype.kind == "f" else np.int32))
        if len({a.dtype for a in flat}) > 1:
            flat = [a.astype(np.float32) for a in flat]
        dt = _hip.DT_F32 if flat[0].dtype == np.float32 else _hip.DT_I32
        ptrs = (ctypes.c_void_p * db.B)(*[a.ctypes.data for a in flat])
        with torch.cuda.device(self.device):
            scores, logits, sp, lp = self._outputs(db, want_logits)
            _hip.check(self.L.mdf_engine_forward_dense(self.handle, db.packed.plan, ctypes.byref(db.desc), ptrs, dt, sp, lp, self._stream()))
        return (scores, logits) if want_logits else scores

    def _dense_view(self, db: DeviceBatch) -> DeviceBatch:
        """`db` re-planned with at most DENSE_CHUNK_ROWS rows per chunk (cached on `db`): same proteins, same order, smaller upload slots."""
        view = getattr(db, "_dense", None)
        if view is None:
            import copy
            pk = copy.copy(db.packed)
            pk._plan(DENSE_CHUNK_ROWS)
            view = db._dense = DeviceBatch(pk, self.device)
        return view

    def lm_features(self, packed: PackedProteins, which: int = 0):
        """Language-model features (LSTM2 output) of every protein of `packed`: list of (L_p, H) float32 arrays.  For
        inspection and tests; the scoring paths keep these on the device."""
        torch = _torch()
        lm = self.lms[which]
        db = self.upload(packed)
        out = np.empty((int(packed.Lq.astype(np.int64).sum()), lm.hidden), dtype=np.float32)
        with torch.cuda.device(self.device):
            _hip.check(self.L.mdf_engine_lm_features_host(self.handle, packed.plan, ctypes.byref(db.desc), int(which), _hip.ptr(out), self._stream()))
        first_invalid_residue(packed, db.bad.cpu().numpy())
        off = np.concatenate(([0], np.cumsum(packed.Lq.astype(np.int64))))
        return [out[off[p]:off[p + 1]].copy() for p in range(packed.B)]

    @property
    def nnz_capacity(self) -> int:
        return int(self.L.mdf_engine_nnz_capacity(self.handle))

    def last_chunk_nnz(self) -> int:
        """CSR entries of the chunk processed last (diagnostic for the A.X roofline of bench.py; synchronises)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            return int(self.L.mdf_engine_last_chunk_nnz(self.handle, self._stream()))

    def graph_stats(self):
        """(forward calls replayed as one hipGraph, forward calls issued launch by launch)"""
        g, e = _hip.c_int64(0), _hip.c_int64(0)
        _hip.check(self.L.mdf_engine_graph_stats(self.handle, g, e))
        return int(g.value), int(e.value)

    def check(self, db: DeviceBatch):
        """Synchronise and raise what the asynchronous stages flagged (invalid residue, CSR overflow)."""
        torch = _torch()
        info = (ctypes.c_int64 * 4)()
        with torch.cuda.device(self.device):
            rc = self.L.mdf_engine_check(self.handle, db.packed.plan, ctypes.byref(db.desc), self._stream(), info)
        if rc == _hip.MDF_EBADCHAR:
            raise ValueError(f"Invalid character in sequence: {db.packed.seqs[info[0]][info[1]]}")
        _hip.check(rc)

    def raise_flags(self, packed: PackedProteins, bad, st):
        """Turn the per-chunk device flags (host copies: bad (n_chunks,) int64, status (n_chunks, 4)) into the exceptions the
        per-call API raises -- for callers that copied the flags themselves (mDeepFRI.stream).  Chunks hold consecutive
        proteins, so the first flagged chunk carries the first invalid byte of the whole batch: what the reference's serial loop
        would have hit first (predict.pyx:36-46)."""
        first_invalid_residue(packed, bad)
        if (st[:, 2] != 0).any():
            raise ValueError(f"a query of length {int(st[:, 2].max())} exceeds the max_len the contact stage was given")
        if (st[:, 0] != 0).any():
            need = int(st[:, 1].max())
            raise _hip.CapacityError(_hip.MDF_ECAPACITY,
                                     f"CSR capacity {self.nnz_capacity} too small (a chunk needs {need}); raise nnz_per_row")

    def run_alignments(self, packed: PackedProteins) -> dict:
        """Convenience: upload, run the fused path, validate, return {mode: np.ndarray (B,T)}.  On a CSR overflow the
        capacity is raised once and the batch re-run."""
        db = self.upload(packed)
        out = self.forward_alignments(db)
        try:
            self.check(db)
        except _hip.CapacityError:
            need = int(db.status.cpu().numpy()[:, 1].max())
            self.nnz_per_row = need // max(packed.max_chunk_rows, 1) + 8
            _hip.check(self.L.mdf_engine_set_nnz_per_row(self.handle, self.nnz_per_row))
            db = self.upload(packed)
            out = self.forward_alignments(db)
            self.check(db)
        return {m: t.cpu().numpy() for m, t in out.items()}


class HostBatch:
    """Host lists of one batch flattened for the library's single-call entries (mdf_engine_run_alignments_host and its pipelined pair): packed
    bytes / floats + per-protein lengths.  What `predict_batch` of the compiled binding builds, in Python."""

    def __init__(self, seqs, coords, q_alns, t_alns):
        if not (len(seqs) == len(coords) == len(q_alns) == len(t_alns)) or len(seqs) == 0:
            raise ValueError("seqs, coords, q_alns and t_alns must be non-empty lists of one length")
        self.seqs = seqs
        self.B = len(seqs)
        self.seq_bytes = "".join(seqs).encode("ascii")
        self.q_bytes, self.t_bytes = "".join(q_alns).encode("ascii"), "".join(t_alns).encode("ascii")
        self.Lq = np.fromiter(map(len, seqs), dtype=np.int32, count=self.B)
        self.La = np.fromiter(map(len, q_alns), dtype=np.int32, count=self.B)
        if not np.array_equal(self.La, np.fromiter(map(len, t_alns), dtype=np.int32, count=self.B)):
            raise ValueError("gapped query and target differ in length")
        cs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 3) for c in coords]
        self.Lt = np.fromiter((c.shape[0] for c in cs), dtype=np.int32, count=self.B)
        self.xyz = np.ascontiguousarray(np.concatenate(cs, axis=0)) if int(self.Lt.sum()) else np.zeros((1, 3), np.float32)


class HostPipeline:
    """Host lists in -> host float32 score arrays out through the library's two-slot pipeline (mdf_engine_submit_alignments_host /
    mdf_engine_collect_host): the batched counterpart of the reference's two loops (pipeline.py:476-481, 292-319) for a caller that has
    Python lists and wants numpy arrays, at the device-resident rate -- packing of batch k + 1 and unpacking of batch k - 1 run under
    the kernels of batch k.

        pipe = HostPipeline(engine)
        for scores in pipe.run(batches):       # batches: iterable of (seqs, coords, q_alns, t_alns); scores: {mode: (B, T) float32}, in order
            ...
    """

    def __init__(self, engine: "HotPathEngine"):
        self.engine = engine
        self.L = engine.L
        self._inflight = []   # (ticket, HostBatch, outputs)

    def submit(self, seqs, coords, q_alns, t_alns):
        hb = seqs if isinstance(seqs, HostBatch) else HostBatch(seqs, coords, q_alns, t_alns)
        eng = self.engine
        out = {m: np.empty((hb.B, p.n_terms), dtype=np.float32) for m, p in eng.predictors.items()}
        ticket = _hip.c_int64(-1)
        with _torch().cuda.device(eng.device):
            _hip.check(self.L.mdf_engine_submit_alignments_host(eng.handle, hb.seq_bytes, _hip.ptr(hb.Lq), hb.B, _hip.ptr(hb.xyz), _hip.ptr(hb.Lt),
                                                                 hb.q_bytes, hb.t_bytes, _hip.ptr(hb.La), ticket))
        self._inflight.append((ticket.value, hb, out))
        return ticket.value

    def collect(self):
        """Scores of the OLDEST batch in flight ({mode: (B, T) float32}); raises what the reference's loop would (ValueError on an invalid residue)."""
        ticket, hb, out = self._inflight.pop(0)
        eng = self.engine
        ptrs = (ctypes.c_void_p * len(eng.modes))(*[out[m].ctypes.data for m in eng.modes])
        info = (_hip.c_int64 * 4)()
        with _torch().cuda.device(eng.device):
            rc = self.L.mdf_engine_collect_host(eng.handle, ticket, ptrs, info)
        if rc == _hip.MDF_EBADCHAR:
            raise ValueError(f"Invalid character in sequence: {hb.seqs[info[0]][info[1]]}")
        _hip.check(rc)
        return out

    def run(self, batches):
        for item in batches:
            if isinstance(item, HostBatch):
                self.submit(item, None, None, None)
            else:
                self.submit(*item)
            if len(self._inflight) == 2:
                yield self.collect()
        while self._inflight:
            yield self.collect()


class SequenceEngine:
    """Sequence-only CNN models for batches of proteins on one GPU: the batched counterpart of the reference's CNN loop
    over the unaligned queries (pipeline.py:600-648, `_run_prediction_loop(predictor=cnn, ...)`).
    `predictors`: {mode: Predictor built from a DeepCNN model}.  A thin caller of the library's `mdf_seq_engine_*`."""

    def __init__(self, predictors: dict, device: int = 0, max_rows: int = 1 << 20):
        import weakref
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("SequenceEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        for m, p in predictors.items():
            if p.session.kind != "cnn":
                raise ValueError(f"predictor {m!r} is not a sequence-only (CNN) model")
        self.L = _hip.lib()
        self.predictors = dict(predictors)
        self.modes = list(self.predictors)
        self.device = torch.device(f"cuda:{device}")
        self.max_rows = int(max_rows)
        handles = (ctypes.c_void_p * len(self.modes))(*[self.predictors[m].session.handle for m in self.modes])
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _hip.check(self.L.mdf_seq_engine_create(handles, len(self.modes), int(device), ctypes.byref(h)))
        self.handle = h
        weakref.finalize(self, self.L.mdf_seq_engine_free, h)

    def forward(self, db: DeviceBatch) -> dict:
        """{mode: (B, T) float32 scores on the device}; asynchronous on the current stream (`check(db)` syncs)."""
        torch = _torch()
        with torch.cuda.device(self.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            out = {m: torch.empty((db.B, p.n_terms), dtype=torch.float32, device=self.device) for m, p in self.predictors.items()}
            sp = (ctypes.c_void_p * len(self.modes))(*[out[m].data_ptr() for m in self.modes])
            _hip.check(self.L.mdf_seq_engine_forward(self.handle, db.packed.plan, ctypes.byref(db.desc), sp, st))
            return out

    def check(self, db: DeviceBatch):
        torch = _torch()
        info = (ctypes.c_int64 * 4)()
        with torch.cuda.device(self.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            rc = self.L.mdf_seq_engine_check(self.handle, db.packed.plan, ctypes.byref(db.desc), st, info)
        if rc == _hip.MDF_EBADCHAR:
            raise ValueError(f"Invalid character in sequence: {db.packed.seqs[info[0]][info[1]]}")
        _hip.check(rc)

    def run(self, seqs) -> dict:
        """Convenience: pack, upload, run, validate -> {mode: np.ndarray (B, T)}."""
        db = DeviceBatch(PackedProteins.pack(seqs, max_rows=self.max_rows), self.device)
        out = self.forward(db)
        self.check(db)
        return {m: t.cpu().numpy() for m, t in out.items()}


# ---------------------------------------------------------------------------------------------------------------------
# batched build_align_contact_map (reference output format)
# ---------------------------------------------------------------------------------------------------------------------
def build_align_contact_maps(alignments, threshold: float = 6, generated_contacts: int = 2, device: int = 0,
                             max_rows: int = 32768):
    """Batched counterpart of `Pool(threads).map(build_align_contact_map, alignments)` (reference pipeline.py:476-481).
    Returns [(alignment, int32 (Lq,Lq) | None), ...] in input order, None (with a warning) where coords is None."""
    import logging
    torch = _torch()
    logger = logging.getLogger("mDeepFRI.bio_utils")
    alignments = list(alignments)
    results = [None] * len(alignments)
    live = []
    for i, a in enumerate(alignments):
        if a.coords is None:
            logger.warning(f"No coordinates found for {a.target_name}.")
            results[i] = (a, None)
        else:
            live.append(i)
    if not live:
        return results
    L = _hip.lib()
    dev = torch.device(f"cuda:{device}")
    seqs = [alignments[i].gapped_sequence.replace("-", "") for i in live]
    nonempty = [k for k, s in enumerate(seqs) if len(s) > 0]
    for k, s in enumerate(seqs):
        if len(s) == 0:
            results[live[k]] = (alignments[live[k]], np.zeros((0, 0), dtype=np.int32))
    if not nonempty:
        return results
    live = [live[k] for k in nonempty]
    seqs = [seqs[k] for k in nonempty]
    pk = PackedProteins.pack(seqs, [alignments[i].coords for i in live], [alignments[i].gapped_sequence for i in live],
                             [alignments[i].gapped_target for i in live], max_rows=max_rows, keep_order=True)   # (the launches below index the batch by plan position)
    # Per chunk: one launch builds the dense maps on the device; they cross PCIe into one of two pinned buffers on a second,
    # high-priority stream while the next chunk is being built, and a few host threads cut the buffer into the freshly allocated
    # (Lq, Lq) arrays the reference hands out (1 MiB each at L = 512: the copies, not the kernels, are the cost of this format).
    from concurrent.futures import ThreadPoolExecutor
    import os

    def cut(host, off, Lp):
        a = np.empty((Lp, Lp), dtype=np.int32)
        np.copyto(a, host[off:off + Lp * Lp].reshape(Lp, Lp))
        return a

    def finish(job, pool):
        ch, offs, host, ev = job
        ev.synchronize()
        lens = [int(pk.Lq[p]) for p in range(ch.p0, ch.p1)]
        for p, a in zip(range(ch.p0, ch.p1), pool.map(cut, [host] * len(lens), offs.tolist(), lens)):
            results[live[p]] = (alignments[live[p]], a)

    with torch.cuda.device(dev), ThreadPoolExecutor(max_workers=max(1, min(8, os.cpu_count() or 1))) as pool:
        db = DeviceBatch(pk, dev)
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev, priority=-1)
        st = ctypes.c_void_p(main.cuda_stream)
        cws = torch.empty(L.mdf_cmap_workspace_bytes(pk.B, pk.max_chunk_rows, 0), dtype=torch.uint8, device=dev)
        pin, job = [None, None], None
        for ci, ch in enumerate(pk.chunks):
            sizes = pk.Lq[ch.p0:ch.p1].astype(np.int64)**2
            offs = np.zeros(len(sizes), dtype=np.int64)
            np.cumsum(sizes[:-1], out=offs[1:])
            n = int(sizes.sum())
            out = torch.empty(n, dtype=torch.int32, device=dev)
            d_off = torch.from_numpy(offs).to(dev)
            _hip.check(L.mdf_cmap_dense_dev(_p(db.coords), _p(db.coord_off, ch.p0), _p(db.q_aln), _p(db.t_aln), _p(db.aln_off, ch.p0),
                                            _p(db.Lq, ch.p0), _p(db.chunk_row_off, ch.row_off_pos), ch.p1 - ch.p0, ch.rows,
                                            float(threshold), int(generated_contacts), _p(out), _p(d_off), _p(cws), cws.numel(), st))
            built = torch.cuda.Event()
            built.record(main)
            s = ci & 1                      # the buffer chunk ci - 2 used: its arrays were cut out before chunk ci - 1 was launched
            if pin[s] is None or pin[s].numel() < n:
                pin[s] = torch.empty(n + n // 4, dtype=torch.int32, pin_memory=True)
            with torch.cuda.stream(side):
                side.wait_event(built)
                out.record_stream(side)
                pin[s][:n].copy_(out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            if job is not None:
                finish(job, pool)
            job = (ch, offs, pin[s].numpy(), ev)
        if job is not None:
            finish(job, pool)
    return results


def prediction_rows(query_ids, scores: np.ndarray, net_type: str = "gcn"):
    """Rows exactly as reference pipeline.py:318 writes them: [query_id, net_type] + pred_vector.tolist()."""
    return [[qid, net_type] + np.asarray(row, dtype=np.float32).tolist() for qid, row in zip(query_ids, scores)]
