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
    group_base: int = 0    # first 32-row group of this chunk inside its segment's partial-sum array


@dataclass
class Segment:
    """Consecutive chunks whose per-group partial sums share one array and are pooled by one launch per GO head."""
    p0: int
    p1: int
    groups: int            # 32-row groups in the segment
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

    @property
    def B(self) -> int:
        return len(self.seqs)

    @classmethod
    def pack(cls, seqs, coords=None, q_alns=None, t_alns=None, max_rows: int = 32768, max_segment_groups: int = 1 << 19):
        seqs = list(seqs)
        if not seqs:
            raise ValueError("empty batch")
        enc = [s.encode("ascii") for s in seqs]
        Lq = np.array([len(b) for b in enc], dtype=np.int32)
        if (Lq <= 0).any():
            raise ValueError("empty sequence in batch")
        pk = cls(seqs=seqs, Lq=Lq, seq_bytes=np.frombuffer(b"".join(enc), dtype=np.uint8).copy(), seq_off=_offsets(Lq))
        if coords is not None:
            if q_alns is None or t_alns is None or not (len(coords) == len(q_alns) == len(t_alns) == len(seqs)):
                raise ValueError("coords, q_alns and t_alns must all be given, one per sequence")
            cs = []
            for c in coords:
                c = np.asarray(c)
                if c.dtype != np.float32 or c.ndim != 2 or c.shape[1] != 3:
                    raise ValueError("coordinates must be float32 (Lt,3)")
                cs.append(np.ascontiguousarray(c))
            pk.coords = np.concatenate(cs, axis=0) if cs else np.zeros((0, 3), np.float32)
            if pk.coords.shape[0] == 0:
                pk.coords = np.zeros((1, 3), np.float32)
            pk.coord_off = _offsets([c.shape[0] for c in cs])
            qb = [q.encode("ascii") for q in q_alns]
            tb = [t.encode("ascii") for t in t_alns]
            for i, (q, t, s) in enumerate(zip(qb, tb, enc)):
                if len(q) != len(t):
                    raise ValueError(f"protein {i}: gapped query and target differ in length")
                if len(q) - q.count(b"-") != len(s):
                    raise ValueError(f"protein {i}: gapped query does not spell a sequence of length {len(s)}")
            pk.q_aln = np.frombuffer(b"".join(qb), dtype=np.uint8).copy()
            pk.t_aln = np.frombuffer(b"".join(tb), dtype=np.uint8).copy()
            pk.aln_off = _offsets([len(q) for q in qb])
        pk._plan(max_rows, max_segment_groups)
        return pk

    @classmethod
    def from_alignments(cls, alignments, max_rows: int = 32768, max_segment_groups: int = 1 << 19):
        """Pack objects carrying the AlignmentResult attributes the reference's path reads (query_sequence, coords,
        gapped_sequence, gapped_target; reference alignment.py:106-150).  Entries without coordinates are skipped, as
        pipeline.py:485 filters them; returns (packed, kept_indices)."""
        keep = [i for i, a in enumerate(alignments) if a.coords is not None]
        al = [alignments[i] for i in keep]
        pk = cls.pack([a.gapped_sequence.replace("-", "") for a in al], [a.coords for a in al], [a.gapped_sequence for a in al],
                      [a.gapped_target for a in al], max_rows=max_rows, max_segment_groups=max_segment_groups)
        return pk, keep

    @classmethod
    def from_aligned_batch(cls, batch, coords, max_rows: int = 32768, max_segment_groups: int = 1 << 19):
        """Pack the struct-of-arrays output of mDeepFRI.alignment.align_queries_arrays (the GPU aligner) together with the
        targets' C-alpha coordinates -- no AlignmentResult objects, no per-protein string handling: the gapped strings are
        taken as the flat byte arrays the aligner produced.  coords: one float32 (Lt, 3) array per query (its best target's
        trace), None where the structure is missing; returns (packed, kept_indices) like from_alignments."""
        keep = [i for i, c in enumerate(coords) if c is not None]
        if not keep:
            raise ValueError("empty batch")
        off = np.asarray(batch.aln_off, dtype=np.int64)
        ln = (off[1:] - off[:-1])[keep]
        sel = np.repeat(off[:-1][keep] - np.concatenate(([0], np.cumsum(ln)[:-1])), ln) + np.arange(int(ln.sum()), dtype=np.int64)
        q_aln, t_aln = np.ascontiguousarray(batch.q_aln[sel]), np.ascontiguousarray(batch.t_aln[sel])
        seqs = [batch.query_sequences[i] for i in keep]
        Lq = np.array([len(s) for s in seqs], dtype=np.int32)
        if (Lq <= 0).any():
            raise ValueError("empty sequence in batch")
        starts = np.concatenate(([0], np.cumsum(ln)))
        nongap = np.add.reduceat((q_aln != 45).astype(np.int64), starts[:-1]) if len(ln) else np.zeros(0, np.int64)
        if not np.array_equal(nongap, Lq):
            raise ValueError("gapped queries do not spell the query sequences")
        pk = cls(seqs=seqs, Lq=Lq, seq_bytes=np.frombuffer("".join(seqs).encode("ascii"), dtype=np.uint8).copy(), seq_off=_offsets(Lq))
        cs = []
        for i in keep:
            c = np.asarray(coords[i])
            if c.dtype != np.float32 or c.ndim != 2 or c.shape[1] != 3:
                raise ValueError("coordinates must be float32 (Lt,3)")
            cs.append(c)
        pk.coords = np.ascontiguousarray(np.concatenate(cs, axis=0))
        pk.coord_off = _offsets([c.shape[0] for c in cs])
        pk.q_aln, pk.t_aln, pk.aln_off = q_aln, t_aln, _offsets(ln)
        pk._plan(max_rows, max_segment_groups)
        return pk, keep

    def _plan(self, max_rows: int, max_segment_groups: int = 1 << 19):
        L = _hip.lib()
        self.chunks, offs = [], []
        p0, B = 0, self.B
        pad = (self.Lq.astype(np.int64) + 31) // 32 * 32
        while p0 < B:
            p1, rows = p0, 0
            while p1 < B and (p1 == p0 or rows + pad[p1] <= max_rows):
                rows += pad[p1]
                p1 += 1
            ro = np.zeros(p1 - p0 + 1, dtype=np.int32)
            lq = np.ascontiguousarray(self.Lq[p0:p1])
            R = L.mdf_layout_rows(_hip.ptr(lq), p1 - p0, _hip.ptr(ro))
            if R < 0:
                _hip.check(int(R))
            self.chunks.append(Chunk(p0, p1, int(R), sum(len(o) for o in offs)))
            offs.append(ro)
            p0 = p1
        self.chunk_row_off = np.concatenate(offs)
        # pooling segments: protein p's 32-row groups are [grp_off[p], grp_off[p+1]) inside its segment's partial array
        self.segments, goffs = [], []
        cur, seg_groups, seg_first = [], 0, 0
        for ci, ch in enumerate(self.chunks):
            g = ch.rows // 32
            if cur and seg_groups + g > max_segment_groups:
                self._close_segment(cur, seg_groups, goffs)
                cur, seg_groups = [], 0
            ch.segment, ch.group_base = len(self.segments), seg_groups
            cur.append(ci)
            seg_groups += g
        self._close_segment(cur, seg_groups, goffs)
        self.grp_off = np.concatenate(goffs)

    def _close_segment(self, chunk_ids, groups, goffs):
        first, last = self.chunks[chunk_ids[0]], self.chunks[chunk_ids[-1]]
        off = np.empty(last.p1 - first.p0 + 1, dtype=np.int32)
        for ci in chunk_ids:
            ch = self.chunks[ci]
            ro = self.chunk_row_off[ch.row_off_pos:ch.row_off_pos + (ch.p1 - ch.p0) + 1]
            off[ch.p0 - first.p0:ch.p1 - first.p0] = ch.group_base + ro[:-1] // 32
        off[-1] = groups
        self.segments.append(Segment(first.p0, last.p1, int(groups), sum(len(o) for o in goffs)))
        goffs.append(off)

    @property
    def max_chunk_rows(self) -> int:
        return max(c.rows for c in self.chunks)


# ---------------------------------------------------------------------------------------------------------------------
# device engine
# ---------------------------------------------------------------------------------------------------------------------
def _p(t, elem_offset: int = 0):
    return ctypes.c_void_p(t.data_ptr() + elem_offset * t.element_size())


class DeviceBatch:
    """PackedProteins uploaded to one GPU (torch tensors)."""

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

    @property
    def B(self):
        return self.packed.B


def first_invalid_residue(packed: PackedProteins, bad) -> None:
    """Raise the reference's ValueError for the lowest (protein, position) flagged by mdf_seq_encode_dev, if any."""
    for ci, ch in enumerate(packed.chunks):
        key = int(bad[ci])
        if key != -1:
            p, pos = ch.p0 + (key >> 32), key & 0xffffffff
            raise ValueError(f"Invalid character in sequence: {packed.seqs[p][pos]}")


class HotPathEngine:
    """contact map + GCN forward for batches of proteins on one GPU, for one or several GO heads
    (`predictors`: {mode: mDeepFRI.predict.Predictor}; all heads share the contact-map stage)."""

    def __init__(self, predictors: dict, device: int = 0, max_rows: int = 32768, nnz_per_row: int = 40,
                 threshold: float = 6.0, generated_contacts: int = 2, lm_batch: int = 8192, lm_workspace_gib: float = 48.0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("HotPathEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.L = _hip.lib()
        self.predictors = dict(predictors)
        self.device = torch.device(f"cuda:{device}")
        self.max_rows = int(max_rows)
        self.nnz_per_row = int(nnz_per_row)
        self.threshold = float(threshold)
        self.generated_contacts = int(generated_contacts)
        self._rows_alloc = self._len_alloc = 0
        self._bufs = {}
        # heads with a language-model branch, grouped by the (shared) LanguageModel they are attached to; the LSTM runs
        # once per group over `lm_batch` proteins at a time (every time step is one GEMM over all of them)
        self.lms = []
        for p in self.predictors.values():
            lm = getattr(p.session, "lm", None)
            if lm is not None and all(lm is not x for x in self.lms):
                self.lms.append(lm)
        self.lm_batch = int(lm_batch)
        self.lm_workspace_bytes = int(lm_workspace_gib * 2**30)

    # -- memory ------------------------------------------------------------------------------------------------------
    def _ensure(self, rows: int, n_proteins: int, max_len: int = 0):
        torch = _torch()
        max_len = (int(max_len) + 63) // 64 * 64
        if rows > self._rows_alloc or max_len > self._len_alloc:
            rows, max_len = max(rows, self._rows_alloc), max(max_len, self._len_alloc)
            dev = self.device
            cap = rows * self.nnz_per_row
            gws = max(self.L.mdf_gcn_workspace_bytes(p.session.handle, rows) for p in self.predictors.values())
            self._bufs = {
                "rowptr": torch.empty(rows + 1, dtype=torch.int32, device=dev),
                "colidx": torch.empty(cap, dtype=torch.int32, device=dev),
                "val": torch.empty(cap, dtype=torch.float32, device=dev),
                "seq_idx": torch.empty(rows, dtype=torch.uint8, device=dev),
                "lsum": torch.empty(rows * 32, dtype=torch.float32, device=dev),
                "cws": torch.empty(self.L.mdf_cmap_workspace_bytes(1 << 20, rows, max_len), dtype=torch.uint8, device=dev),
                "gws": torch.empty(gws, dtype=torch.uint8, device=dev),
            }
            self._rows_alloc, self._nnz_cap, self._len_alloc = rows, cap, max_len
        hws = max(self.L.mdf_head_workspace_bytes(p.session.handle, n_proteins) for p in self.predictors.values())
        if self._bufs.get("hws") is None or self._bufs["hws"].numel() < hws:
            self._bufs["hws"] = torch.empty(hws, dtype=torch.uint8, device=self.device)

    def upload(self, packed: PackedProteins) -> DeviceBatch:
        return DeviceBatch(packed, self.device)

    def _stream(self):
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    # -- stages ------------------------------------------------------------------------------------------------------
    def _gcn_chunk(self, db: DeviceBatch, ch: Chunk, partial: dict, st, seq_ptr, lm_h=None, have_lsum: bool = False):
        """letter sums once per chunk (shared by every head without a language model), then the GraphConv stack of each
        head; the per-group partial sums land in the head's segment array.  `lm_h`: {LanguageModel: pointer to this
        chunk's (rows, H) language-model features}."""
        b = self._bufs
        if not have_lsum and any(getattr(p.session, "lm", None) is None for p in self.predictors.values()):
            _hip.check(self.L.mdf_letter_sums_dev(seq_ptr, _p(b["rowptr"]), _p(b["colidx"]), _p(b["val"]), ch.rows, _p(b["lsum"]), st))
        for mode, pred in self.predictors.items():
            feat = pred.session.topology["feature_dim"]
            lm = getattr(pred.session, "lm", None)
            if lm is None:
                _hip.check(self.L.mdf_gcn_embed_dev(pred.session.handle, _p(b["lsum"]), _p(b["rowptr"]), _p(b["colidx"]), _p(b["val"]),
                                                    ch.rows, _p(partial[mode], ch.group_base * feat), _p(b["gws"]), b["gws"].numel(), st))
            else:
                _hip.check(self.L.mdf_gcn_embed_lm_dev(pred.session.handle, seq_ptr, lm_h[id(lm)], _p(b["rowptr"]), _p(b["colidx"]),
                                                       _p(b["val"]), ch.rows, _p(partial[mode], ch.group_base * feat), _p(b["gws"]),
                                                       b["gws"].numel(), st))

    # -- language model ------------------------------------------------------------------------------------------------
    def _lm_batches(self, packed: PackedProteins):
        """Consecutive chunk ranges [c0, c1) whose proteins run through the LSTM together: at most `lm_batch` proteins and
        a time-major workspace (2 x (Lmax+1) x B x H floats) within `lm_workspace_bytes`.  The proteins are dealt evenly
        over the fewest such groups: an LSTM time step costs whole rounds of 256x256 tiles, so a small trailing group
        would cost as much as a full one."""
        H = max(lm.hidden for lm in self.lms)
        cap = min(self.lm_batch, 65535)
        n_groups = max(1, -(-packed.B // cap))
        while True:
            target = -(-packed.B // n_groups)
            out, c0, nb, lmax, ok = [], 0, 0, 0, True
            for ci, ch in enumerate(packed.chunks):
                n = ch.p1 - ch.p0
                l = int(packed.Lq[ch.p0:ch.p1].max())
                if ci > c0 and (nb + n > cap or nb >= target or 8 * (max(lmax, l) + 1) * (nb + n) * H > self.lm_workspace_bytes):
                    out.append((c0, ci))
                    c0, nb, lmax = ci, 0, 0
                nb, lmax = nb + n, max(lmax, l)
            out.append((c0, len(packed.chunks)))
            if len(out) <= n_groups or n_groups >= len(packed.chunks):
                return out
            n_groups = len(out)   # memory or chunk granularity forced more groups: re-balance for that count

    def _lm_forward(self, db: DeviceBatch, c0: int, c1: int, bases, seq_all, st):
        """LSTM features of every protein in chunks [c0, c1) -> {id(lm): (rows_total, H) tensor} in residue-row layout."""
        torch = _torch()
        pk = db.packed
        chunks = pk.chunks[c0:c1]
        rows_total = bases[-1]
        prot_row = np.concatenate([bases[k] + pk.chunk_row_off[ch.row_off_pos:ch.row_off_pos + (ch.p1 - ch.p0)].astype(np.int64)
                                   for k, ch in enumerate(chunks)])
        lens = pk.Lq[chunks[0].p0:chunks[-1].p1]
        order = np.argsort(-lens.astype(np.int64), kind="stable")
        lens_h = np.ascontiguousarray(lens[order], dtype=np.int32)
        d_rows = torch.from_numpy(np.ascontiguousarray(prot_row[order])).to(self.device)
        d_lens = torch.from_numpy(lens_h).to(self.device)
        B, Lmax = len(lens_h), int(lens_h[0])
        out = {}
        for lm in self.lms:
            need = self.L.mdf_lm_workspace_bytes(lm.handle, B, Lmax)
            if self._bufs.get("lm_ws") is None or self._bufs["lm_ws"].numel() < need:
                self._bufs["lm_ws"] = None
                self._bufs["lm_ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
            key = ("lm_h", id(lm))
            if self._bufs.get(key) is None or self._bufs[key].numel() < rows_total * lm.hidden:
                self._bufs[key] = None
                self._bufs[key] = torch.zeros(rows_total * lm.hidden, dtype=torch.float32, device=self.device)
            _hip.check(self.L.mdf_lm_forward_dev(lm.handle, _p(seq_all), _p(d_rows), _p(d_lens), _hip.ptr(lens_h), B, _p(self._bufs[key]),
                                                 _p(self._bufs["lm_ws"]), self._bufs["lm_ws"].numel(), st))
            out[id(lm)] = self._bufs[key]
        self._keep = (d_rows, d_lens, lens_h)  # outlive the asynchronous launches
        return out

    def _pool_segment(self, db: DeviceBatch, seg: Segment, partial: dict, pooled: dict, st):
        for mode, pred in self.predictors.items():
            feat = pred.session.topology["feature_dim"]
            _hip.check(self.L.mdf_gcn_pool_dev(pred.session.handle, _p(partial[mode]), _p(db.grp_off, seg.grp_off_pos), seg.p1 - seg.p0,
                                               _p(pooled[mode], seg.p0 * feat), st))

    def _alloc_partial(self, db):
        torch = _torch()
        g = max(sg.groups for sg in db.packed.segments)
        key = ("partial", g)
        if self._bufs.get("partial_key") != key:
            self._bufs["partial"] = {m: torch.empty(g * p.session.topology["feature_dim"], dtype=torch.float32, device=self.device)
                                     for m, p in self.predictors.items()}
            self._bufs["partial_key"] = key
        return self._bufs["partial"]

    def lm_features(self, packed: PackedProteins, which: int = 0):
        """Language-model features (LSTM2 output) of every protein of `packed`: list of (L_p, H) float32 arrays.  For
        inspection and tests; the scoring paths keep these on the device."""
        torch = _torch()
        lm = self.lms[which]
        db = self.upload(packed)
        res = []
        with torch.cuda.device(self.device):
            st = self._stream()
            keep_lms, self.lms = self.lms, [lm]
            try:
                for c0, c1 in self._lm_batches(packed):
                    chunks = packed.chunks[c0:c1]
                    bases = [0]
                    for ch in chunks:
                        bases.append(bases[-1] + ch.rows)
                    seq_all = torch.empty(bases[-1], dtype=torch.uint8, device=self.device)
                    for k, ch in enumerate(chunks):
                        _hip.check(self.L.mdf_seq_encode_dev(_p(db.seq_bytes), _p(db.seq_off, ch.p0), _p(db.Lq, ch.p0),
                                                             _p(db.chunk_row_off, ch.row_off_pos), ch.p1 - ch.p0, ch.rows,
                                                             _p(seq_all, bases[k]), _p(db.bad, c0 + k), st))
                    feats = self._lm_forward(db, c0, c1, bases, seq_all, st)[id(lm)]
                    torch.cuda.current_stream(self.device).synchronize()
                    host = feats[:bases[-1] * lm.hidden].view(bases[-1], lm.hidden).cpu().numpy()
                    for k, ch in enumerate(chunks):
                        ro = packed.chunk_row_off[ch.row_off_pos:ch.row_off_pos + (ch.p1 - ch.p0)]
                        for j, p in enumerate(range(ch.p0, ch.p1)):
                            r0 = bases[k] + int(ro[j])
                            res.append(host[r0:r0 + int(packed.Lq[p])].copy())
            finally:
                self.lms = keep_lms
        return res

    def _run_chunks(self, db: DeviceBatch, encode, build_csr, st):
        """Common driver: per chunk `encode(ci, ch, seq_ptr)` writes the residue indices and `build_csr(ci, ch, seq_ptr)` the
        adjacency (returning True when it also produced the layer-1 letter sums), then the GCN stack runs; segments are pooled as soon as their last chunk has been issued.  With a
        language model the chunks are taken `lm_batch` proteins at a time: all of them are encoded first, the LSTM runs
        over the whole group, then the per-chunk stages follow."""
        torch = _torch()
        pooled, partial = self._alloc_pooled(db), self._alloc_partial(db)
        chunks, segs = db.packed.chunks, db.packed.segments
        b = self._bufs

        def tail(ci, ch, seq_ptr, lm_h):
            have_lsum = bool(build_csr(ci, ch, seq_ptr))
            self._gcn_chunk(db, ch, partial, st, seq_ptr, lm_h, have_lsum)
            if ci + 1 == len(chunks) or chunks[ci + 1].segment != ch.segment:
                self._pool_segment(db, segs[ch.segment], partial, pooled, st)

        if not self.lms:
            for ci, ch in enumerate(chunks):
                encode(ci, ch, _p(b["seq_idx"]))
                tail(ci, ch, _p(b["seq_idx"]), None)
            return pooled
        for c0, c1 in self._lm_batches(db.packed):
            bases = [0]
            for ch in chunks[c0:c1]:
                bases.append(bases[-1] + ch.rows)
            if b.get("seq_all") is None or b["seq_all"].numel() < bases[-1]:
                b["seq_all"] = torch.empty(bases[-1], dtype=torch.uint8, device=self.device)
            for k, ci in enumerate(range(c0, c1)):
                encode(ci, chunks[ci], _p(b["seq_all"], bases[k]))
            feats = self._lm_forward(db, c0, c1, bases, b["seq_all"], st)
            hid = {id(lm): lm.hidden for lm in self.lms}
            for k, ci in enumerate(range(c0, c1)):
                lm_h = {key: _p(t, bases[k] * hid[key]) for key, t in feats.items()}
                tail(ci, chunks[ci], _p(b["seq_all"], bases[k]), lm_h)
        return pooled

    def _heads(self, db: DeviceBatch, pooled: dict, want_logits: bool, st):
        torch = _torch()
        scores, logits = {}, {}
        for mode, pred in self.predictors.items():
            T = pred.n_terms
            scores[mode] = torch.empty((db.B, T), dtype=torch.float32, device=self.device)
            lg = torch.empty((db.B, 2 * T), dtype=torch.float32, device=self.device) if want_logits else None
            _hip.check(self.L.mdf_gcn_head_dev(pred.session.handle, _p(pooled[mode]), db.B, _p(scores[mode]),
                                               _p(lg) if lg is not None else None, _p(self._bufs["hws"]),
                                               self._bufs["hws"].numel(), st))
            if want_logits:
                logits[mode] = lg
        return (scores, logits) if want_logits else scores

    def _alloc_pooled(self, db):
        torch = _torch()
        return {m: torch.empty((db.B, p.session.topology["feature_dim"]), dtype=torch.float32, device=self.device)
                for m, p in self.predictors.items()}

    def forward_alignments(self, db: DeviceBatch, want_logits: bool = False):
        """Fused path: coords + alignments + sequences -> {mode: (B,T) float32 scores on the device}.  Asynchronous
        on the current stream; call `check(db)` (one sync) before trusting the result."""
        if db.coords is None:
            raise ValueError("batch was packed without coordinates/alignments")
        torch = _torch()
        with torch.cuda.device(self.device):
            max_len = int(db.packed.Lq.max())
            self._ensure(db.packed.max_chunk_rows, db.B, max_len)
            b, st = self._bufs, self._stream()
            want_lsum = any(getattr(p.session, "lm", None) is None for p in self.predictors.values())

            def encode(ci, ch, seq_ptr):
                _hip.check(self.L.mdf_seq_encode_dev(_p(db.seq_bytes), _p(db.seq_off, ch.p0), _p(db.Lq, ch.p0),
                                                     _p(db.chunk_row_off, ch.row_off_pos), ch.p1 - ch.p0, ch.rows, seq_ptr,
                                                     _p(db.bad, ci), st))

            def build_csr(ci, ch, seq_ptr):
                # contact stage: coordinates read once; the CSR fill also writes the layer-1 letter sums of the chunk
                Bc = ch.p1 - ch.p0
                ro = _p(db.chunk_row_off, ch.row_off_pos)
                _hip.check(self.L.mdf_cmap_csr_dev(
                    _p(db.coords), _p(db.coord_off, ch.p0), _p(db.q_aln), _p(db.t_aln), _p(db.aln_off, ch.p0), _p(db.Lq, ch.p0), ro,
                    Bc, ch.rows, max_len, self.threshold, self.generated_contacts, _p(b["rowptr"]), _p(b["colidx"]), _p(b["val"]),
                    self._nnz_cap, _p(db.status, ci * 4), seq_ptr if want_lsum else None, _p(b["lsum"]) if want_lsum else None,
                    _p(b["cws"]), b["cws"].numel(), st))
                return want_lsum

            pooled = self._run_chunks(db, encode, build_csr, st)
            return self._heads(db, pooled, want_logits, st)

    def forward_dense(self, db: DeviceBatch, cmaps, want_logits: bool = False):
        """Reference-format path: one dense (L,L) contact map per protein (what build_align_contact_map returns,
        int32) -> scores.  Maps are uploaded chunk by chunk."""
        torch = _torch()
        if len(cmaps) != db.B:
            raise ValueError("one contact map per protein expected")
        with torch.cuda.device(self.device):
            self._ensure(db.packed.max_chunk_rows, db.B)
            b, st = self._bufs, self._stream()
            keep = []  # device copies of the maps must outlive the asynchronous kernels that read them

            def encode(ci, ch, seq_ptr):
                _hip.check(self.L.mdf_seq_encode_dev(_p(db.seq_bytes), _p(db.seq_off, ch.p0), _p(db.Lq, ch.p0),
                                                     _p(db.chunk_row_off, ch.row_off_pos), ch.p1 - ch.p0, ch.rows, seq_ptr,
                                                     _p(db.bad, ci), st))

            def build_csr(ci, ch, seq_ptr):
                Bc = ch.p1 - ch.p0
                ro = _p(db.chunk_row_off, ch.row_off_pos)
                flat, offs = [], [0]
                for p in range(ch.p0, ch.p1):
                    A = np.asarray(cmaps[p])
                    Lp = int(db.packed.Lq[p])
                    if A.shape != (Lp, Lp):
                        raise ValueError(f"protein {p}: cmap shape {A.shape} != ({Lp},{Lp})")
                    flat.append(np.ascontiguousarray(A, dtype=np.float32 if A.dtype.kind == "f" else np.int32).reshape(-1))
                    offs.append(offs[-1] + Lp * Lp)
                if len({a.dtype for a in flat}) > 1:
                    flat = [a.astype(np.float32) for a in flat]
                host = np.concatenate(flat)
                d_maps = torch.from_numpy(host).to(self.device)
                d_off = torch.from_numpy(np.asarray(offs[:-1], dtype=np.int64)).to(self.device)
                keep.append((d_maps, d_off))
                nnz_needed = int(sum(int(np.count_nonzero(a)) for a in flat)) + ch.rows
                if nnz_needed > self._nnz_cap:
                    torch.cuda.current_stream(self.device).synchronize()
                    b["colidx"] = torch.empty(nnz_needed, dtype=torch.int32, device=self.device)
                    b["val"] = torch.empty(nnz_needed, dtype=torch.float32, device=self.device)
                    self._nnz_cap = nnz_needed
                dt = _hip.DT_F32 if host.dtype == np.float32 else _hip.DT_I32
                _hip.check(self.L.mdf_dense_to_csr_dev(_p(d_maps), dt, _p(d_off), _p(db.Lq, ch.p0), ro, Bc, ch.rows, _p(b["rowptr"]),
                                                       _p(b["colidx"]), _p(b["val"]), self._nnz_cap, _p(db.status, ci * 4),
                                                       _p(b["cws"]), b["cws"].numel(), st))
                if len(keep) > 2:  # bound the device memory held by uploaded maps
                    torch.cuda.current_stream(self.device).synchronize()
                    del keep[:-1]

            pooled = self._run_chunks(db, encode, build_csr, st)
            out = self._heads(db, pooled, want_logits, st)
            torch.cuda.current_stream(self.device).synchronize()
            return out

    def check(self, db: DeviceBatch):
        """Synchronise and raise what the asynchronous stages flagged (invalid residue, CSR overflow)."""
        torch = _torch()
        torch.cuda.current_stream(self.device).synchronize()
        self.raise_flags(db.packed, db.bad.cpu().numpy(), db.status.cpu().numpy())

    def raise_flags(self, packed: PackedProteins, bad, st):
        """Turn the per-chunk device flags (host copies: bad (n_chunks,) int64, status (n_chunks, 4)) into the exceptions the
        per-call API raises.  Chunks hold consecutive proteins, so the first flagged chunk carries the first invalid byte of
        the whole batch: what the reference's serial loop would have hit first (predict.pyx:36-46)."""
        first_invalid_residue(packed, bad)
        if (st[:, 2] != 0).any():
            raise ValueError(f"a query of length {int(st[:, 2].max())} exceeds the max_len the contact stage was given")
        if (st[:, 0] != 0).any():
            need = int(st[:, 1].max())
            raise _hip.CapacityError(_hip.MDF_ECAPACITY,
                                     f"CSR capacity {self._nnz_cap} too small (a chunk needs {need}); raise nnz_per_row")

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
            self._rows_alloc = 0
            db = self.upload(packed)
            out = self.forward_alignments(db)
            self.check(db)
        return {m: t.cpu().numpy() for m, t in out.items()}


class SequenceEngine:
    """Sequence-only CNN models for batches of proteins on one GPU: the batched counterpart of the reference's CNN loop
    over the unaligned queries (pipeline.py:600-648, `_run_prediction_loop(predictor=cnn, ...)`).
    `predictors`: {mode: Predictor built from a DeepCNN model}."""

    def __init__(self, predictors: dict, device: int = 0, max_rows: int = 1 << 20):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("SequenceEngine needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        for m, p in predictors.items():
            if p.session.kind != "cnn":
                raise ValueError(f"predictor {m!r} is not a sequence-only (CNN) model")
        self.L = _hip.lib()
        self.predictors = dict(predictors)
        self.device = torch.device(f"cuda:{device}")
        self.max_rows = int(max_rows)

    def forward(self, db: DeviceBatch) -> dict:
        """{mode: (B, T) float32 scores on the device}; asynchronous on the current stream (`check(db)` syncs)."""
        torch = _torch()
        pk = db.packed
        with torch.cuda.device(self.device):
            st = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            rows = pk.max_chunk_rows
            seq_idx = torch.empty(rows, dtype=torch.uint8, device=self.device)
            ws = torch.empty((rows // 32 + 1) * 4 + 512, dtype=torch.uint8, device=self.device)
            out = {m: torch.empty((db.B, p.n_terms), dtype=torch.float32, device=self.device) for m, p in self.predictors.items()}
            # conv + max pool chunk by chunk into one (B, C) array per head; the output layer then runs ONCE per head over all
            # proteins (a 2 048-protein chunk is too few rows to fill the GEMM's 256-row tiles on 256 CUs)
            cpad = {m: int(self.L.mdf_cnn_padded_channels(p.session.handle)) for m, p in self.predictors.items()}
            pooled = {m: torch.empty((db.B, cpad[m]), dtype=torch.float32, device=self.device) for m in self.predictors}
            for ci, ch in enumerate(pk.chunks):
                Bc = ch.p1 - ch.p0
                ro = _p(db.chunk_row_off, ch.row_off_pos)
                _hip.check(self.L.mdf_seq_encode_dev(_p(db.seq_bytes), _p(db.seq_off, ch.p0), _p(db.Lq, ch.p0), ro, Bc, ch.rows,
                                                     _p(seq_idx), _p(db.bad, ci), st))
                for m, p in self.predictors.items():
                    _hip.check(self.L.mdf_cnn_pool_dev(p.session.handle, _p(seq_idx), _p(db.Lq, ch.p0), ro, Bc, ch.rows,
                                                       _p(pooled[m], ch.p0 * cpad[m]), _p(ws), ws.numel(), st))
            for m, p in self.predictors.items():
                _hip.check(self.L.mdf_cnn_head_dev(p.session.handle, _p(pooled[m]), db.B, _p(out[m]), st))
            self._keep = (seq_idx, ws, pooled)
            return out

    def check(self, db: DeviceBatch):
        torch = _torch()
        torch.cuda.current_stream(self.device).synchronize()
        first_invalid_residue(db.packed, db.bad.cpu().numpy())

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
                             [alignments[i].gapped_target for i in live], max_rows=max_rows)
    with torch.cuda.device(dev):
        db = DeviceBatch(pk, dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        cws = torch.empty(L.mdf_cmap_workspace_bytes(pk.B, pk.max_chunk_rows, 0), dtype=torch.uint8, device=dev)
        for ch in pk.chunks:
            sizes = pk.Lq[ch.p0:ch.p1].astype(np.int64)**2
            offs = np.zeros(len(sizes), dtype=np.int64)
            np.cumsum(sizes[:-1], out=offs[1:])
            out = torch.empty(int(sizes.sum()), dtype=torch.int32, device=dev)
            d_off = torch.from_numpy(offs).to(dev)
            _hip.check(L.mdf_cmap_dense_dev(_p(db.coords), _p(db.coord_off, ch.p0), _p(db.q_aln), _p(db.t_aln), _p(db.aln_off, ch.p0),
                                            _p(db.Lq, ch.p0), _p(db.chunk_row_off, ch.row_off_pos), ch.p1 - ch.p0, ch.rows,
                                            float(threshold), int(generated_contacts), _p(out), _p(d_off), _p(cws), cws.numel(), st))
            host = out.cpu().numpy()
            for k, p in enumerate(range(ch.p0, ch.p1)):
                Lp = int(pk.Lq[p])
                results[live[p]] = (alignments[live[p]], host[offs[k]:offs[k] + Lp * Lp].reshape(Lp, Lp).copy())
    return results


def prediction_rows(query_ids, scores: np.ndarray, net_type: str = "gcn"):
    """Rows exactly as reference pipeline.py:318 writes them: [query_id, net_type] + pred_vector.tolist()."""
    return [[qid, net_type] + np.asarray(row, dtype=np.float32).tolist() for qid, row in zip(query_ids, scores)]
