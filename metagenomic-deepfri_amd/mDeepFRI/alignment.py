"""Input side of the hot path, with the public names of the reference's mDeepFRI/alignment.py: `insert_gaps` (:38-62), the
`AlignmentResult` carrier (:65-150), and the alignment step itself -- `best_hit_database` (:164-196), `align_pairwise`
(:198-221), `pairwise_against_database` (:223-250) -- with PyOpal's SIMD Needleman-Wunsch replaced by the batched HIP kernels
of csrc/nw.hip (one wave per pair).  `align_queries` is the batched counterpart of the reference's
`Pool(threads).starmap(pairwise_against_database, ...)` (:266-320): ONE score launch over every (query, candidate) pair, the
per-query arg-max, ONE full-alignment launch over the winners.

Scoring matrices: the reference takes "VTML80" from the `scoring_matrices` package.  Neither that package nor any copy of the
table exists offline, so a matrix is an INPUT here: `ScoringMatrix.from_name` asks `scoring_matrices` when it is importable,
`ScoringMatrix.from_file` reads the NCBI text format, `ScoringMatrix.simple` builds match/mismatch tables.  Parity with PyOpal
itself is unpinned (oracle/nw_oracle.c states the model and the tie rules used)."""
from __future__ import annotations

from typing import Optional, Tuple

import ctypes
import threading
from itertools import islice

import numpy as np

from . import _hip


def insert_gaps(sequence: str, reference: str, alignment_string: str) -> Tuple[str, str]:
    """Gapped query / target strings from an alignment string: column i marked 'I' puts '-' into the query at index i,
    'D' puts '-' into the target at index i, every other letter ('M', 'X', ...) leaves both untouched
    (reference alignment.py:38-62; an index past the current end appends, as list.insert does)."""
    q, t = bytearray(sequence, "ascii"), bytearray(reference, "ascii")
    for i, a in enumerate(alignment_string):
        if a == "I":
            q[min(i, len(q)):min(i, len(q))] = b"-"
        elif a == "D":
            t[min(i, len(t)):min(i, len(t))] = b"-"
    return q.decode("ascii"), t.decode("ascii")


class AlignmentResult:
    """Carrier with the attribute names the path reads (reference alignment.py:106-150): query_name, query_sequence,
    target_name, target_sequence, alignment, identities/coverages, db_name, coords, gapped_sequence, gapped_target."""

    def __init__(self, query_name: str = "", query_sequence: str = "", target_name: str = "", target_sequence: str = "",
                 alignment: str = "", query_identity: Optional[float] = None, query_coverage: Optional[float] = None,
                 target_coverage: Optional[float] = None, db_name: Optional[str] = None, coords: Optional[np.ndarray] = None):
        self.query_name = query_name
        self.query_sequence = query_sequence
        self.target_name = target_name
        self.target_sequence = target_sequence
        self.alignment = alignment
        self.query_identity = query_identity
        self.query_coverage = query_coverage
        self.target_coverage = target_coverage
        self.insert_gaps()
        self.db_name = db_name
        self.coords = coords
        self.target_coords = None
        self.cmap = None
        self.aligned_cmap = None

    def insert_gaps(self):
        self.gapped_sequence, self.gapped_target = insert_gaps(self.query_sequence, self.target_sequence, self.alignment)

    def __repr__(self):
        return (f"AlignmentResult(query_name={self.query_name}, target_name={self.target_name}, "
                f"query_identity={self.query_identity}, query_coverage={self.query_coverage})")

    __str__ = __repr__


# ---------------------------------------------------------------------------------------------------------------------
# scoring matrices
# ---------------------------------------------------------------------------------------------------------------------
class ScoringMatrix:
    """Substitution scores over an alphabet: `alphabet` (str, <= 32 letters) and `matrix` (int32 (A, A), matrix[q][t])."""

    def __init__(self, alphabet: str, matrix, name: str = "custom"):
        self.alphabet = str(alphabet)
        self.matrix = np.ascontiguousarray(matrix, dtype=np.int32)
        self.name = name
        A = len(self.alphabet)
        if not (0 < A <= 32) or self.matrix.shape != (A, A) or len(set(self.alphabet)) != A:
            raise ValueError("scoring matrix must be (A, A) over an alphabet of A <= 32 distinct letters")
        self._lut = np.full(256, 255, dtype=np.uint8)
        for i, c in enumerate(self.alphabet):
            self._lut[ord(c)] = i
        self._lut_nocase = self._lut.copy()      # the batched entry upper-cases on the device: a lower-case letter is its capital
        for i, c in enumerate(self.alphabet):
            if c.isalpha() and self._lut_nocase[ord(c.lower())] == 255:
                self._lut_nocase[ord(c.lower())] = i

    @classmethod
    def simple(cls, alphabet: str = "ARNDCQEGHILKMFPSTWYVBZX*", match: int = 5, mismatch: int = -4):
        A = len(alphabet)
        m = np.full((A, A), mismatch, dtype=np.int32)
        np.fill_diagonal(m, match)
        return cls(alphabet, m, name=f"simple({match},{mismatch})")

    @classmethod
    def from_file(cls, path: str):
        """NCBI / EMBOSS text format: '#' comments, a header row of letters, then one row per letter."""
        rows, header = [], None
        with open(path) as f:
            for line in f:
                line = line.strip()
                if not line or line.startswith("#"):
                    continue
                tok = line.split()
                if header is None:
                    header = tok
                    continue
                rows.append((tok[0], [int(float(x)) for x in tok[1:1 + len(header)]]))
        if header is None or [r[0] for r in rows] != header:
            raise ValueError(f"{path}: not a square substitution matrix in NCBI format")
        return cls("".join(header), np.array([r[1] for r in rows], dtype=np.int32), name=path)

    @classmethod
    def from_name(cls, name: str):
        """The reference's `ScoringMatrix.from_name(name)` (alignment.py:175): served by the `scoring_matrices` package when it
        is installed; there is no built-in copy of VTML80 (no source for the table exists in this build's environment)."""
        try:
            from scoring_matrices import ScoringMatrix as _SM
        except ImportError:
            raise ImportError(f"scoring matrix {name!r}: the `scoring_matrices` package is not installed and no table is built in; "
                              f"pass a ScoringMatrix (ScoringMatrix.from_file(path) reads the NCBI text format)") from None
        sm = _SM.from_name(name)
        return cls(sm.alphabet, np.array([list(r) for r in sm.matrix]).round().astype(np.int32), name=name)

    def encode(self, seq: str) -> np.ndarray:
        codes = self._lut[np.frombuffer(seq.encode("ascii"), dtype=np.uint8)]
        if (codes == 255).any():
            raise ValueError(f"character {seq[int(np.argmax(codes == 255))]!r} is not in the scoring matrix alphabet")
        return codes


def _matrix(scoring_matrix) -> ScoringMatrix:
    return scoring_matrix if isinstance(scoring_matrix, ScoringMatrix) else ScoringMatrix.from_name(scoring_matrix)


def _upper(seq):
    return seq.upper() if seq else seq


class _PairBatch:
    """Encoded sequences + pair lists in the layout of mdf_nw_* (include/mdfri.h)."""

    def __init__(self, sequences, sm: ScoringMatrix):
        """sequences: list of str, any case (the reference upper-cases every sequence it aligns, alignment.py:152-161); encoded in
        one pass over their concatenation."""
        self.seq_len = np.fromiter(map(len, sequences), dtype=np.int32, count=len(sequences))
        self.seq_off = np.zeros(len(sequences), dtype=np.int64)
        if len(sequences) > 1:
            np.cumsum(self.seq_len[:-1], out=self.seq_off[1:])
        raw = np.frombuffer("".join(sequences).upper().encode("ascii"), dtype=np.uint8)
        self.codes = sm._lut[raw] if raw.size else np.zeros(1, np.uint8)
        if raw.size and (self.codes == 255).any():
            raise ValueError(f"character {chr(int(raw[int(np.argmax(self.codes == 255))]))!r} is not in the scoring matrix alphabet")
        self.sm = sm

    def _by_cost(self, pq, pt):
        """Launch order: largest DP matrices first.  One wave sweeps one pair, so a 2 000 x 2 000 pair started last would run on
        alone for milliseconds after everything else has finished (longest-processing-time-first scheduling)."""
        return np.argsort(-(self.seq_len[pq].astype(np.int64) * self.seq_len[pt].astype(np.int64)), kind="stable")

    def scores(self, pair_q, pair_t, gap_open, gap_extend) -> np.ndarray:
        pq, pt = np.asarray(pair_q, dtype=np.int32), np.asarray(pair_t, dtype=np.int32)
        out = np.empty(len(pq), dtype=np.int32)
        if len(pq):
            order = self._by_cost(pq, pt)
            pq, pt = np.ascontiguousarray(pq[order]), np.ascontiguousarray(pt[order])
            sc = np.empty(len(pq), dtype=np.int32)
            _hip.check(_hip.lib().mdf_nw_score_host(_hip.ptr(self.codes), _hip.ptr(self.seq_off), _hip.ptr(self.seq_len), len(self.seq_len),
                                                   _hip.ptr(pq), _hip.ptr(pt), len(pq), _hip.ptr(self.sm.matrix), len(self.sm.alphabet),
                                                   int(gap_open), int(gap_extend), _hip.ptr(sc)))
            out[order] = sc
        return out

    def align(self, pair_q, pair_t, gap_open, gap_extend, max_trace_bytes: int = 512 << 20, tie_rule: int = 0):
        """Full alignments -> dict of arrays: ops / q_aln / t_aln (flat uint8), off (P+1, int64: pair p's columns are
        [off[p], off[p+1])), op_len, n_match, score.  Pairs are processed in groups whose direction bytes fit `max_trace_bytes`."""
        L = _hip.lib()
        pq0, pt0 = np.asarray(pair_q, dtype=np.int32), np.asarray(pair_t, dtype=np.int32)
        P = len(pq0)
        order = self._by_cost(pq0, pt0) if P else np.zeros(0, np.int64)
        pq, pt = np.ascontiguousarray(pq0[order]), np.ascontiguousarray(pt0[order])
        parts = []
        cost = (self.seq_len[pt].astype(np.int64) + 63) // 64 * ((self.seq_len[pq].astype(np.int64) + 66) // 4 * 4) * 64   # = mdf_nw_plan's trace bytes
        p0 = 0
        while p0 < P:
            p1, acc = p0, 0
            while p1 < P and (p1 == p0 or acc + cost[p1] <= max_trace_bytes):
                acc += cost[p1]
                p1 += 1
            n = p1 - p0
            q_, t_ = np.ascontiguousarray(pq[p0:p1]), np.ascontiguousarray(pt[p0:p1])
            ops_off = np.zeros(n + 1, dtype=np.int64)
            _hip.check(L.mdf_nw_plan(_hip.ptr(self.seq_len), _hip.ptr(q_), _hip.ptr(t_), n, None, None, _hip.ptr(ops_off)))
            cap = max(int(ops_off[-1]), 1)
            ops, qa, ta = (np.empty(cap, dtype=np.uint8) for _ in range(3))
            op_len, n_match, score = (np.empty(n, dtype=np.int32) for _ in range(3))
            _hip.check(L.mdf_nw_align_host(_hip.ptr(self.codes), _hip.ptr(self.seq_off), _hip.ptr(self.seq_len), len(self.seq_len), _hip.ptr(q_),
                                           _hip.ptr(t_), n, _hip.ptr(self.sm.matrix), len(self.sm.alphabet), int(gap_open), int(gap_extend), int(tie_rule),
                                           self.sm.alphabet.encode("ascii"), _hip.ptr(ops), _hip.ptr(qa), _hip.ptr(ta), _hip.ptr(op_len),
                                           _hip.ptr(n_match), _hip.ptr(score)))
            # compact: every pair's columns sit right-aligned in its capacity [ops_off[p], ops_off[p+1])
            start = ops_off[1:] - op_len
            off = np.zeros(n + 1, dtype=np.int64)
            np.cumsum(op_len, out=off[1:])
            idx = np.repeat(start - off[:-1], op_len) + np.arange(int(off[-1]), dtype=np.int64)
            parts.append((ops[idx], qa[idx], ta[idx], op_len, n_match, score))
            p0 = p1
        cat = lambda k, dt: np.concatenate([p[k] for p in parts]) if parts else np.zeros(0, dt)  # noqa: E731
        # back to the caller's pair order: per-pair arrays by inverse permutation, the flat column arrays by a gather
        len_s = cat(3, np.int32)
        off_s = np.zeros(P + 1, dtype=np.int64)
        np.cumsum(len_s, out=off_s[1:])
        inv = np.empty(P, dtype=np.int64)
        inv[order] = np.arange(P)
        op_len = len_s[inv]
        off = np.zeros(P + 1, dtype=np.int64)
        np.cumsum(op_len, out=off[1:])
        src = np.repeat(off_s[:-1][inv] - off[:-1], op_len) + np.arange(int(off[-1]), dtype=np.int64)
        return {"ops": cat(0, np.uint8)[src], "q_aln": cat(1, np.uint8)[src], "t_aln": cat(2, np.uint8)[src], "off": off, "op_len": op_len,
                "n_match": cat(4, np.int32)[inv], "score": cat(5, np.int32)[inv]}


def _identity(n_match, op_len):
    """pyopal FullResult.identity(): matches / alignment length, a C float there -- float32 here, then a Python float."""
    return float(np.float32(n_match) / np.float32(max(int(op_len), 1)))


def best_hit_database(query, target_sequences, gap_open: int = 10, gap_extend: int = 1, scoring_matrix="VTML80"):
    """reference alignment.py:164-196: (key, sequence) of the candidate with the highest global alignment score; the first one
    wins a tie (Python's max over results in database order)."""
    sm = _matrix(scoring_matrix)
    query = _upper(query)
    keys = list(target_sequences)
    targets = [_upper(target_sequences[k]) for k in keys]
    pb = _PairBatch([query] + targets, sm)
    sc = pb.scores(np.zeros(len(keys), np.int32), np.arange(1, len(keys) + 1, dtype=np.int32), gap_open, gap_extend)
    best = int(np.argmax(sc))
    return keys[best], targets[best]


MAX_TRACE_BYTES = 512 << 20   # device memory the direction words of one alignment launch may take (align_queries_arrays)
TIE_RULE = 0   # which co-optimal alignment is returned (3 bits, see include/mdfri.h / oracle/nw_oracle.c); PyOpal's choice is unpinned


def align_pairwise(query, target, gap_open: int = 10, gap_extend: int = 1, scoring_matrix="VTML80", tie_rule: int | None = None):
    """reference alignment.py:198-221 -> (alignment string, identity, query coverage, target coverage)."""
    sm = _matrix(scoring_matrix)
    pb = _PairBatch([_upper(query), _upper(target)], sm)
    r = pb.align([0], [1], gap_open, gap_extend, tie_rule=TIE_RULE if tie_rule is None else tie_rule)
    return bytes(r["ops"]).decode(), _identity(r["n_match"][0], r["op_len"][0]), 1.0, 1.0   # a global alignment covers both sequences


def pairwise_against_database(query_id, query_sequence, target_sequences, gap_open: int = 10, gap_extend: int = 1,
                              scoring_matrix="VTML80") -> AlignmentResult:
    """reference alignment.py:223-250: best hit of one query among its candidates + the alignment with it -- the batched entry with a
    batch of one (a single library call instead of a score call and an alignment call)."""
    return align_queries_arrays([query_id], [query_sequence], [target_sequences], gap_open, gap_extend, scoring_matrix).results()[0]


class AlignedBatch:
    """Struct-of-arrays result of `align_queries_arrays`: no per-protein Python objects.  Protein p's gapped strings are
    q_aln[aln_off[p]:aln_off[p+1]] / t_aln[...]; `best` is the index of the winning candidate inside that query's list."""

    def __init__(self, query_ids, query_sequences, target_keys, target_sequences, best, res):
        self.query_ids, self.query_sequences, self.target_keys, self.target_sequences = query_ids, query_sequences, target_keys, target_sequences
        self.best = best
        self.ops, self.q_aln, self.t_aln, self.aln_off = res["ops"], res["q_aln"], res["t_aln"], res["off"]
        self.op_len, self.n_match, self.score = res["op_len"], res["n_match"], res["score"]
        self.identity = (self.n_match.astype(np.float32) / np.maximum(self.op_len, 1).astype(np.float32))

    def __len__(self):
        return len(self.query_ids)

    def results(self):
        """AlignmentResult objects, as the reference's pool returns them (alignment.py:313-320)."""
        out = []
        for p in range(len(self)):
            a, b = int(self.aln_off[p]), int(self.aln_off[p + 1])
            r = AlignmentResult.__new__(AlignmentResult)
            AlignmentResult.__init__(r, self.query_ids[p], self.query_sequences[p], self.target_keys[p], self.target_sequences[p], "",
                                     float(self.identity[p]), query_coverage=1.0, target_coverage=1.0)
            r.alignment = bytes(self.ops[a:b]).decode()
            r.gapped_sequence, r.gapped_target = bytes(self.q_aln[a:b]).decode(), bytes(self.t_aln[a:b]).decode()
            out.append(r)
        return out


class AlignerWorkspace:
    """What a long-lived caller of the batched aligner keeps between calls (`mdf_nw_workspace`: a stream, device scratch, pinned
    staging, the state of one call in flight) on one device; freed with the object.  `stream`: a HIP stream handle (int) the aligner's
    launches go to -- they then sit in that stream's order --, None for a stream of the workspace's own.  Without a workspace a
    call uses the calling thread's own, which lives as long as the process."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        import weakref
        L = _hip.lib()
        h = ctypes.c_void_p()
        _hip.check(L.mdf_nw_workspace_create(int(device), ctypes.c_void_p(stream) if stream else None, ctypes.byref(h)))
        self.handle, self.device = h, int(device)
        weakref.finalize(self, L.mdf_nw_workspace_free, h)


class PendingAlignment:
    """A batch of queries on its way through the aligner (`mdf_nw_best_hits_begin` / `_align` / `_finish`): built by
    `align_queries_begin` (scores enqueued), `launch_alignments()` waits for the scores and enqueues the winners' alignments,
    `result()` waits for those and returns the AlignedBatch.  Each wait is for work enqueued one step earlier."""

    def __init__(self, workspace, query_ids, seqs, nq, target_sequences, first, cand, cap, joined):
        self.workspace, self.query_ids, self.seqs, self.nq = workspace, query_ids, seqs, nq
        self.target_sequences, self.first, self.cand, self.cap, self.joined = target_sequences, first, cand, cap, joined
        self.stage = 1

    def _fail(self, rc, info):
        self.stage = 0
        if rc == _hip.MDF_EBADCHAR:
            raise ValueError(f"character {self.seqs[int(info[0])][int(info[1])]!r} is not in the scoring matrix alphabet")
        _hip.check(rc)

    def launch_alignments(self):
        if self.stage != 1:
            raise RuntimeError("launch_alignments: the scores of this batch are not in flight")
        info = np.zeros(4, dtype=np.int64)
        rc = _hip.lib().mdf_nw_best_hits_align(self.workspace.handle, _hip.ptr(info))
        if rc:
            self._fail(rc, info)
        self.stage = 2
        return self

    def result(self) -> "AlignedBatch":
        if self.stage == 1:
            self.launch_alignments()
        if self.stage != 2:
            raise RuntimeError("result: no alignments in flight for this batch")
        nq, cap = self.nq, self.cap
        ops, qa, ta = (np.empty(cap, dtype=np.uint8) for _ in range(3))
        best, score, op_len, n_match = (np.empty(nq, dtype=np.int32) for _ in range(4))
        off = np.empty(nq + 1, dtype=np.int64)
        info = np.zeros(4, dtype=np.int64)
        rc = _hip.lib().mdf_nw_best_hits_finish(self.workspace.handle, _hip.ptr(best), _hip.ptr(score), _hip.ptr(op_len), _hip.ptr(n_match), _hip.ptr(off),
                                                _hip.ptr(ops), _hip.ptr(qa), _hip.ptr(ta), cap, None, _hip.ptr(info))
        if rc:
            self._fail(rc, info)
        self.stage = 0
        n = int(off[-1])
        seqs, joined = self.seqs, self.joined
        # the reference upper-cases every sequence it aligns (alignment.py:152-161); the usual input already is
        upper = seqs if joined.isupper() or not joined else "\n".join(seqs).upper().split("\n")
        bt = self.cand[self.first[:-1] + best]
        res = {"ops": ops[:n], "q_aln": qa[:n], "t_aln": ta[:n], "off": off, "op_len": op_len, "n_match": n_match, "score": score}
        return AlignedBatch(self.query_ids, upper[:nq], [next(islice(d, b, None)) for d, b in zip(self.target_sequences, best.tolist())],
                            [upper[j] for j in bt.tolist()], best.astype(np.int64), res)

    def abandon(self):
        if self.stage:
            _hip.lib().mdf_nw_best_hits_abandon(self.workspace.handle)
            self.stage = 0


_thread_state = threading.local()


def _thread_workspace() -> AlignerWorkspace:
    """The calling thread's own workspace on its current device (freed when the thread ends)."""
    dev = _hip.current_device()
    if dev < 0:
        raise _hip.MdfriError(_hip.MDF_ENODEVICE, "no HIP device: the aligner has no CPU fallback")
    table = _thread_state.__dict__.setdefault("workspaces", {})
    if dev not in table:
        table[dev] = AlignerWorkspace(dev)
    return table[dev]


def align_queries_begin(query_ids, query_sequences, target_sequences, gap_open: int = 10, gap_extend: int = 1, scoring_matrix="VTML80",
                        workspace: Optional[AlignerWorkspace] = None) -> Optional[PendingAlignment]:
    """First step of `align_queries_arrays`: stage the sequences and ENQUEUE the score launch over every (query, candidate) pair and
    the arg-max per query.  Returns a PendingAlignment (None for an empty batch); one batch in flight per workspace."""
    sm = _matrix(scoring_matrix)
    query_ids = list(query_ids)
    seqs = list(query_sequences)
    nq = len(seqs)
    if nq == 0:
        return None
    target_sequences = list(target_sequences)
    first = np.zeros(nq + 1, dtype=np.int64)
    np.cumsum(np.fromiter(map(len, target_sequences), dtype=np.int64, count=nq), out=first[1:])
    if (first[1:] == first[:-1]).any():
        raise ValueError("every query needs at least one candidate target")
    # unique targets travel once (the same database entry is a candidate of many queries): sequence index = nq + first-seen rank
    flat = [t for d in target_sequences for t in d.values()]
    uniq = dict.fromkeys(flat)
    for i, t in enumerate(uniq):
        uniq[t] = nq + i
    cand = np.fromiter(map(uniq.__getitem__, flat), dtype=np.int32, count=len(flat))
    seqs.extend(uniq)
    seq_len = np.fromiter(map(len, seqs), dtype=np.int32, count=len(seqs))
    seq_off = np.zeros(len(seqs), dtype=np.int64)
    np.cumsum(seq_len[:-1], out=seq_off[1:])
    joined = "".join(seqs)
    text = np.frombuffer(joined.encode("ascii"), dtype=np.uint8)
    if text.size == 0:
        text = np.zeros(1, np.uint8)
    cap = max(int(seq_len[:nq].sum(dtype=np.int64) + np.maximum.reduceat(seq_len[cand], first[:-1]).sum(dtype=np.int64)), 1)
    ws = workspace if workspace is not None else _thread_workspace()
    _hip.check(_hip.lib().mdf_nw_best_hits_begin(ws.handle, _hip.ptr(text), _hip.ptr(seq_off), _hip.ptr(seq_len), len(seqs), _hip.ptr(sm._lut_nocase), nq,
                                                _hip.ptr(cand), _hip.ptr(first), _hip.ptr(sm.matrix), len(sm.alphabet), int(gap_open), int(gap_extend),
                                                int(TIE_RULE), sm.alphabet.encode("ascii"), int(MAX_TRACE_BYTES), 0))
    return PendingAlignment(ws, query_ids, seqs, nq, target_sequences, first, cand, cap, joined)


def align_queries_arrays(query_ids, query_sequences, target_sequences, gap_open: int = 10, gap_extend: int = 1, scoring_matrix="VTML80",
                         workspace: Optional[AlignerWorkspace] = None) -> AlignedBatch:
    """Batched `pairwise_against_database` over many queries (reference alignment.py:266-320: one pool task per query).
    target_sequences: one {key: sequence} dict per query (its candidate set).  Everything between the host lists and the host arrays
    happens in the library (`mdf_nw_best_hits_*`): a score launch over all candidates, the arg-max per query, alignment launches over
    the winners, packing in query order; queries with an empty candidate set are not allowed (the reference never builds them)."""
    pending = align_queries_begin(query_ids, query_sequences, target_sequences, gap_open, gap_extend, scoring_matrix, workspace)
    if pending is None:
        z = np.zeros(0, np.int32)
        return AlignedBatch([], [], [], [], np.zeros(0, np.int64), {"ops": np.zeros(0, np.uint8), "q_aln": np.zeros(0, np.uint8), "t_aln": np.zeros(0, np.uint8),
                                                                   "off": np.zeros(1, np.int64), "op_len": z, "n_match": z, "score": z})
    try:
        return pending.result()
    except BaseException:
        pending.abandon()
        raise


def align_queries(query_ids, query_sequences, target_sequences, gap_open: int = 10, gap_extend: int = 1, scoring_matrix="VTML80"):
    """List of AlignmentResult, the return value of reference align_mmseqs_results' pool (alignment.py:313-320)."""
    return align_queries_arrays(query_ids, query_sequences, target_sequences, gap_open, gap_extend, scoring_matrix).results()
