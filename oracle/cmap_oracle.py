"""ctypes front-end for oracle/cmap_oracle.c  --  TEST INFRASTRUCTURE, never the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Function names mirror the reference symbols they check (file:line in cmap_oracle.c).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcmap_oracle.so")
_lib = None

_i64 = ctypes.c_int64
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build():
    src = os.path.join(_HERE, "cmap_oracle.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libcmap_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_pairwise_sqeuclidean_f32.argtypes = [_f32p, _i64, _i64, _f32p]
        L.orc_pairwise_sqeuclidean_f32.restype = None
        L.orc_threshold_sq_f32.argtypes = [ctypes.c_double]
        L.orc_threshold_sq_f32.restype = ctypes.c_float
        L.orc_contacts_lt_i32.argtypes = [_f32p, _i64, ctypes.c_float, _i32p]
        L.orc_contacts_lt_i32.restype = None
        L.orc_argwhere_eq1_i32.argtypes = [_i32p, _i64, _i32p]
        L.orc_argwhere_eq1_i32.restype = _i64
        L.orc_align_len.argtypes = [ctypes.c_char_p, ctypes.c_char_p, _i64]
        L.orc_align_len.restype = _i64
        L.orc_align_contact_map.argtypes = [ctypes.c_char_p, ctypes.c_char_p, _i64, _i32p, _i64,
                                            ctypes.c_int, _i32p]
        L.orc_align_contact_map.restype = ctypes.c_int
        L.orc_build_align_contact_map.argtypes = [_f32p, _i64, ctypes.c_char_p, ctypes.c_char_p, _i64,
                                                  ctypes.c_double, ctypes.c_int, _i32p]
        L.orc_build_align_contact_map.restype = ctypes.c_int
        L.orc_seq2onehot.argtypes = [ctypes.c_char_p, _i64, _f32p]
        L.orc_seq2onehot.restype = _i64
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _ip(a):
    return a.ctypes.data_as(_i32p)


def pairwise_sqeuclidean(X):
    X = np.ascontiguousarray(X, dtype=np.float32)
    n, m = X.shape
    D = np.empty((n, n), dtype=np.float32)
    lib().orc_pairwise_sqeuclidean_f32(_fp(X), n, m, _fp(D))
    return D


def contacts_lt(D, threshold):
    """(D < threshold**2).astype(int32), threshold given un-squared as in bio_utils.calculate_contact_map."""
    D = np.ascontiguousarray(D, dtype=np.float32)
    n = D.shape[0]
    out = np.empty((n, n), dtype=np.int32)
    L = lib()
    L.orc_contacts_lt_i32(_fp(D), n, L.orc_threshold_sq_f32(float(threshold)), _ip(out))
    return out


def argwhere_eq1(cmap):
    cmap = np.ascontiguousarray(cmap, dtype=np.int32)
    n = cmap.shape[0]
    cnt = lib().orc_argwhere_eq1_i32(_ip(cmap), n, None)
    pairs = np.empty((cnt, 2), dtype=np.int32)
    lib().orc_argwhere_eq1_i32(_ip(cmap), n, _ip(pairs))
    return pairs


def calculate_contact_map(coords, threshold=6.0, mode="matrix"):
    cm = contacts_lt(pairwise_sqeuclidean(coords), threshold)
    return argwhere_eq1(cm) if mode == "sparse" else cm


def align_contact_map(q, t, pairs, generated_contacts=2):
    qb, tb = q.encode("ascii"), t.encode("ascii")
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    Lq = lib().orc_align_len(qb, tb, len(qb))
    out = np.empty((Lq, Lq), dtype=np.int32)
    rc = lib().orc_align_contact_map(qb, tb, len(qb), _ip(pairs), pairs.shape[0], int(generated_contacts),
                                     _ip(out))
    assert rc == 0
    return out


def build_align_contact_map(coords, q, t, threshold=6.0, generated_contacts=2):
    qb, tb = q.encode("ascii"), t.encode("ascii")
    coords = np.ascontiguousarray(coords, dtype=np.float32)
    Lq = lib().orc_align_len(qb, tb, len(qb))
    out = np.empty((Lq, Lq), dtype=np.int32)
    rc = lib().orc_build_align_contact_map(_fp(coords), coords.shape[0], qb, tb, len(qb), float(threshold),
                                           int(generated_contacts), _ip(out))
    assert rc == 0
    return out


def seq2onehot(seq):
    sb = seq.encode("ascii")
    out = np.empty((len(sb), 26), dtype=np.float32)
    bad = lib().orc_seq2onehot(sb, len(sb), _fp(out))
    if bad >= 0:
        raise ValueError(f"Invalid character in sequence: {seq[bad]}")
    return out
