"""ctypes front end of oracle/nw_oracle.c  --  TEST INFRASTRUCTURE, never the product.  PARITY UNPINNED (see the C header:
PyOpal / scoring_matrices are absent offline).  Names mirror reference mDeepFRI/alignment.py:164-250."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libnw_oracle.so")
_lib = None
_u8p, _i32p = ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int32)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "nw_oracle.c")
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE, "libnw_oracle.so"])
        L = ctypes.CDLL(_SO)
        common = [_u8p, ctypes.c_int32, _u8p, ctypes.c_int32, _i32p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]
        L.nwo_score.argtypes, L.nwo_score.restype = common, ctypes.c_int32
        L.nwo_align.argtypes, L.nwo_align.restype = common + [ctypes.c_int32, ctypes.c_char_p, _i32p, _i32p], ctypes.c_int32
        L.nwo_score_of_ops.argtypes, L.nwo_score_of_ops.restype = common + [ctypes.c_char_p, ctypes.c_int32], ctypes.c_int32
        _lib = L
    return _lib


def encode(seq: str, alphabet: str) -> np.ndarray:
    """Residue letters -> matrix indices (pyopal.Database(..., alphabet=...) does the same; unknown letters are an error)."""
    lut = np.full(256, 255, dtype=np.uint8)
    for i, c in enumerate(alphabet):
        lut[ord(c)] = i
    codes = lut[np.frombuffer(seq.upper().encode("ascii"), dtype=np.uint8)]
    if (codes == 255).any():
        bad = seq[int(np.argmax(codes == 255))]
        raise ValueError(f"character {bad!r} is not in the scoring matrix alphabet")
    return np.ascontiguousarray(codes)


def _args(q, t, matrix, alphabet, gap_open, gap_extend):
    qc, tc = encode(q, alphabet), encode(t, alphabet)
    S = np.ascontiguousarray(matrix, dtype=np.int32)
    return qc, tc, S, (qc.ctypes.data_as(_u8p), len(qc), tc.ctypes.data_as(_u8p), len(tc), S.ctypes.data_as(_i32p), S.shape[0], int(gap_open), int(gap_extend))


def nw_score(q, t, matrix, alphabet, gap_open=10, gap_extend=1) -> int:
    *keep, a = _args(q, t, matrix, alphabet, gap_open, gap_extend)
    return int(lib().nwo_score(*a))


def align_pairwise(q, t, matrix, alphabet, gap_open=10, gap_extend=1, tie_rule=0):
    """alignment.py:198-221 -> (alignment string, identity, query coverage, target coverage); plus the score."""
    *keep, a = _args(q, t, matrix, alphabet, gap_open, gap_extend)
    ops = ctypes.create_string_buffer(len(q) + len(t) + 1)
    score, nm = ctypes.c_int32(0), ctypes.c_int32(0)
    n = lib().nwo_align(*a, int(tie_rule), ops, ctypes.byref(score), ctypes.byref(nm))
    s = ops.raw[:n].decode()
    identity = float(np.float32(nm.value) / np.float32(max(n, 1)))
    return s, identity, 1.0, 1.0, int(score.value)


def score_of_alignment(q, t, ops, matrix, alphabet, gap_open=10, gap_extend=1) -> int:
    *keep, a = _args(q, t, matrix, alphabet, gap_open, gap_extend)
    return int(lib().nwo_score_of_ops(*a, ops.encode(), len(ops)))


def best_hit_database(query, targets: dict, matrix, alphabet, gap_open=10, gap_extend=1):
    """alignment.py:164-196: key and sequence of the target with the highest NW score; the FIRST one on a tie (Python's max)."""
    keys = list(targets)
    scores = [nw_score(query, targets[k], matrix, alphabet, gap_open, gap_extend) for k in keys]
    best = int(np.argmax(scores))      # argmax returns the first maximum, as max(results, key=score) does
    return keys[best], targets[keys[best]].upper()
