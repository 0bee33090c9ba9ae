# cython: language_level=3
"""The binding a reference maintainer adds to mDeepFRI/contact_map_utils.pyx (INTEGRATION.md section B), as a real, compiled
Cython module: same `cpdef` signatures as the reference (contact_map_utils.pyx:17,44 -- including the `threads` keyword the
.pyi omits), bodies replaced by calls into include/mdfri.h / libmdfri_hip.so.  Builder-authored; no reference code."""
import numpy as np

cimport numpy as cnp
from libc.stdint cimport int32_t, int64_t

cnp.import_array()

cdef extern from "mdfri.h":
    const char *mdf_last_error()
    int mdf_pairwise_sqeuclidean_f32(const float *X, int64_t n, int64_t m, float *D, int threads) nogil
    int mdf_align_len(const char *q, const char *t, int64_t La, int64_t *Lq) nogil
    int mdf_align_contact_map(const char *q, const char *t, int64_t La, const int32_t *pairs, int64_t N,
                              int generated_contacts, int32_t *out, int threads) nogil


cdef int _check(int rc) except -1:
    if rc != 0:
        msg = mdf_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise RuntimeError(f"libmdfri_hip error {rc}: {msg}")
    return 0


cpdef pairwise_sqeuclidean(float[:, ::1] X, int threads=1):
    cdef Py_ssize_t n = X.shape[0], m = X.shape[1]
    cdef cnp.ndarray[cnp.float32_t, ndim=2, mode="c"] D = np.empty((n, n), dtype=np.float32)
    cdef int rc
    cdef const float *xp = &X[0, 0] if n > 0 and m > 0 else NULL
    cdef float *dp = <float *>D.data if n > 0 else NULL
    with nogil:
        rc = mdf_pairwise_sqeuclidean_f32(xp, n, m, dp, threads)
    _check(rc)
    return D


cpdef cnp.ndarray[cnp.int32_t, ndim=2] align_contact_map(str query_alignment, str target_alignment,
                                                         cnp.ndarray[cnp.int32_t, ndim=2] sparse_target_contact_map,
                                                         int generated_contacts=2, int threads=1):
    cdef bytes q = query_alignment.encode("ascii")
    cdef bytes t = target_alignment.encode("ascii")
    if len(t) < len(q):
        raise ValueError("target_alignment is shorter than query_alignment")
    cdef cnp.ndarray[cnp.int32_t, ndim=2, mode="c"] pairs
    if sparse_target_contact_map.shape[0] > 0:
        pairs = np.ascontiguousarray(sparse_target_contact_map[:, :2])
    else:
        pairs = np.zeros((0, 2), dtype=np.int32)
    cdef int64_t Lq = 0
    cdef const char *qp = q
    cdef const char *tp = t
    cdef int64_t La = len(q), N = pairs.shape[0]
    _check(mdf_align_len(qp, tp, La, &Lq))
    cdef cnp.ndarray[cnp.int32_t, ndim=2, mode="c"] out = np.empty((Lq, Lq), dtype=np.int32)
    cdef int rc
    cdef const int32_t *pp = <const int32_t *>pairs.data
    cdef int32_t *op = <int32_t *>out.data
    with nogil:
        rc = mdf_align_contact_map(qp, tp, La, pp, N, generated_contacts, op, threads)
    _check(rc)
    return out
