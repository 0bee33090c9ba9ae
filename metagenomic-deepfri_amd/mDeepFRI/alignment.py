"""Input side of the hot path: the two pieces of the reference's mDeepFRI/alignment.py that the path consumes --
`insert_gaps` (alignment.py:38-62) and the `AlignmentResult` attribute contract (alignment.py:65-150) -- without the
PyOpal aligner itself (out of scope: SURVEY.md section 2 row 8).  Lets callers build the batch API's inputs from
(query, target, alignment string) triples exactly as the reference would."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


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
