"""Drop-in for the reference's Cython module mDeepFRI/contact_map_utils.pyx (same names, arguments, dtypes,
exceptions), computing on the GPU through libmdfri_hip.so."""
from __future__ import annotations

import numpy as np

from . import _hip

__all__ = ["pairwise_sqeuclidean", "align_contact_map"]


def _require_buffer(a, dtype, ndim, c_contig, what):
    # Mirrors the checks Cython performs when binding `float[:, ::1]` / `ndarray[int32, ndim=2]` arguments:
    # wrong dtype / rank / contiguity -> ValueError, non-array -> TypeError.
    if not isinstance(a, np.ndarray):
        try:
            a = np.asarray(memoryview(a))
        except TypeError:
            raise TypeError(f"{what}: expected a NumPy array, got {type(a).__name__}")
    if a.dtype != dtype:
        raise ValueError(f"Buffer dtype mismatch, expected '{np.dtype(dtype).name}' but got '{a.dtype.name}'")
    if a.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions (expected {ndim}, got {a.ndim})")
    if c_contig and not a.flags.c_contiguous:
        raise ValueError("ndarray is not C-contiguous")
    return a


def pairwise_sqeuclidean(X, threads: int = 1) -> np.ndarray:
    """reference contact_map_utils.pyx:17-37.  X: float32 (n,m) C-contiguous -> float32 (n,n), bit-exact."""
    X = _require_buffer(X, np.float32, 2, True, "X")
    n, m = X.shape
    D = np.empty((n, n), dtype=np.float32)
    _hip.check(_hip.lib().mdf_pairwise_sqeuclidean_f32(_hip.ptr(X), n, m, _hip.ptr(D), int(threads)))
    return D


def align_contact_map(query_alignment: str, target_alignment: str, sparse_target_contact_map,
                      generated_contacts: int = 2, threads: int = 1) -> np.ndarray:
    """reference contact_map_utils.pyx:44-117.  Returns int32 (Lq,Lq), bit-exact with the reference."""
    if not isinstance(query_alignment, str) or not isinstance(target_alignment, str):
        raise TypeError("Argument 'query_alignment'/'target_alignment' has incorrect type (expected str)")
    q = query_alignment.encode("ascii")
    t = target_alignment.encode("ascii")
    pairs = _require_buffer(sparse_target_contact_map, np.int32, 2, False, "sparse_target_contact_map")
    if pairs.shape[0] and pairs.shape[1] < 2:
        raise ValueError("sparse_target_contact_map must have two columns")
    pairs = np.ascontiguousarray(pairs[:, :2]) if pairs.shape[0] else np.zeros((0, 2), np.int32)
    if len(t) < len(q):
        # the reference reads t_ptr[i] for i < len(q) unchecked (undefined behaviour); refuse instead
        raise ValueError("target_alignment is shorter than query_alignment")
    L = _hip.lib()
    lq = _hip.c_int64(0)
    _hip.check(L.mdf_align_len(q, t, len(q), lq))
    out = np.empty((lq.value, lq.value), dtype=np.int32)
    _hip.check(L.mdf_align_contact_map(q, t, len(q), _hip.ptr(pairs), pairs.shape[0], int(generated_contacts),
                                       _hip.ptr(out), int(threads)))
    return out
