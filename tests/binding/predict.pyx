# cython: language_level=3
"""The binding a reference maintainer adds to mDeepFRI/predict.pyx (INTEGRATION.md section B), compiled: `seq2onehot` and the
`cdef class Predictor` with the reference's public attributes (predict.pyx:50-60) and methods, the onnxruntime session replaced by
the model handles of include/mdfri.h.  The weight tensors come from the pure-Python container readers of this build
(mDeepFRI.weights: .mdfw / .npz / .onnx) and cross the boundary as plain float pointers (mdf_model_create / mdf_cnn_create /
mdf_lm_create).  Builder-authored; no reference code."""
import numpy as np

cimport numpy as cnp
from libc.stdint cimport int32_t, int64_t
from libc.stdlib cimport free, malloc

cnp.import_array()

cdef extern from "mdfri.h":
    ctypedef struct mdf_model
    ctypedef struct mdf_lm
    ctypedef struct mdf_cnn
    ctypedef struct mdf_gcn_weights:
        int32_t embed
        int32_t n_gc
        int32_t gc_dims[3]
        int32_t fc_dim
        int32_t n_terms
        const float *W_aa
        const float *W_gc[3]
        const float *W_fc
        const float *b_fc
        const float *W_out
        const float *b_out
        int32_t lm_dim
        const float *W_lm
        const float *b_lm
        int32_t embed_linear
    ctypedef struct mdf_lm_weights:
        int32_t hidden
        const float *W1
        const float *U1
        const float *b1
        const float *W2
        const float *U2
        const float *b2
    ctypedef struct mdf_cnn_weights:
        int32_t n_branch
        const int32_t *kernel_len
        const int32_t *filters
        const int32_t *pad_left
        const float *const *W
        const float *const *b
        const float *bn_gamma
        const float *bn_beta
        const float *bn_mean
        const float *bn_var
        float bn_eps
        int32_t n_terms
        const float *W_out
        const float *b_out
    const char *mdf_last_error()
    int mdf_seq2onehot(const char *seq, int64_t L, float *out, int64_t *bad_idx) nogil
    int mdf_model_create(const mdf_gcn_weights *w, int device, mdf_model **out)
    void mdf_model_free(mdf_model *m)
    int mdf_model_num_terms(const mdf_model *m)
    int mdf_lm_create(const mdf_lm_weights *w, int device, mdf_lm **out)
    void mdf_lm_free(mdf_lm *lm)
    int mdf_model_attach_lm(mdf_model *m, mdf_lm *lm)
    int mdf_gcn_forward_host(mdf_model *m, const char *seq, int64_t L, const void *cmap, int cmap_dtype, float *scores,
                             int64_t *bad_idx) nogil
    int mdf_cnn_create(const mdf_cnn_weights *w, int device, mdf_cnn **out)
    void mdf_cnn_free(mdf_cnn *m)
    int mdf_cnn_num_terms(const mdf_cnn *m)
    int mdf_cnn_forward_host(mdf_cnn *m, const char *seq, int64_t L, float *scores, int64_t *bad_idx) nogil
    ctypedef struct mdf_engine
    ctypedef struct mdf_engine_config:
        int32_t max_rows
        int32_t nnz_per_row
        double threshold
        int32_t generated_contacts
        int32_t max_segment_groups
        int32_t lm_batch
        double lm_workspace_gib
        int32_t graph_max_chunks
        int32_t pipeline_contact
    int mdf_engine_create(mdf_model *const *models, int32_t n_models, int device, const mdf_engine_config *cfg, mdf_engine **out)
    void mdf_engine_free(mdf_engine *e)
    int mdf_engine_run_alignments_host(mdf_engine *e, const char *seqs, const int32_t *Lq, int32_t B, const float *coords,
                                       const int32_t *Lt, const char *q_aln, const char *t_aln, const int32_t *La,
                                       float *const *scores_host, int64_t *info) nogil
    int mdf_engine_submit_alignments_host(mdf_engine *e, const char *seqs, const int32_t *Lq, int32_t B, const float *coords,
                                          const int32_t *Lt, const char *q_aln, const char *t_aln, const int32_t *La, int64_t *ticket) nogil
    int mdf_engine_collect_host(mdf_engine *e, int64_t ticket, float *const *scores_host, int64_t *info) nogil

DEF MDF_EBADCHAR = -4
DEF MDF_DT_F32 = 1
DEF MAX_HEADS = 8


cdef int _check(int rc) except -1:
    if rc != 0:
        msg = mdf_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise RuntimeError(f"libmdfri_hip error {rc}: {msg}")
    return 0


cdef const float *_fp(cnp.ndarray a):
    return <const float *>a.data


cpdef cnp.ndarray[float, ndim=2] seq2onehot(str seq):
    cdef bytes b = seq.encode("ascii")
    cdef Py_ssize_t L = len(b)
    cdef cnp.ndarray[float, ndim=2, mode="c"] out = np.zeros((L, 26), dtype=np.float32)
    cdef int64_t bad = -1
    cdef int rc = 0
    cdef const char *sp = b
    cdef float *op = <float *>out.data
    if L > 0:
        with nogil:
            rc = mdf_seq2onehot(sp, L, op, &bad)
    if rc == MDF_EBADCHAR:
        raise ValueError(f"Invalid character in sequence: {seq[bad]}")
    _check(rc)
    return out


cdef class Predictor(object):
    cdef public str model_path
    cdef public int threads
    cdef public object session      # attribute kept for compatibility: a small description of what was loaded
    cdef public list input_names
    cdef mdf_model *_gcn
    cdef mdf_lm *_lm
    cdef mdf_cnn *_cnn
    cdef int _T

    def __init__(self, model_path: str, threads: int = 1, ):
        self.model_path = model_path
        self.threads = threads
        self._load_model()

    def _load_model(self):
        from mDeepFRI import weights as W
        w = W.load_weights(W.resolve_model_path(self.model_path))
        arrs = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}
        if W.model_kind(w) == "cnn":
            self._load_cnn(arrs, W.validate_cnn(w))
            self.input_names = ["seq"]
        else:
            self._load_gcn(arrs, W.validate(w))
            self.input_names = ["cmap", "seq"]
        self.session = {"kind": "cnn" if self._cnn != NULL else "gcn", "n_terms": self._T, "device": 0}

    cdef _load_gcn(self, dict a, dict topo):
        cdef mdf_gcn_weights g
        cdef mdf_lm_weights l
        cdef int k
        g.embed, g.n_gc, g.fc_dim, g.n_terms = topo["embed"], len(topo["gc_dims"]), topo["fc_dim"], topo["n_terms"]
        for k in range(3):
            g.gc_dims[k] = topo["gc_dims"][k] if k < g.n_gc else 0
            g.W_gc[k] = _fp(a[f"W_gc{k + 1}"]) if k < g.n_gc else NULL
        g.W_aa, g.W_fc, g.b_fc, g.W_out, g.b_out = _fp(a["W_aa"]), _fp(a["W_fc"]), _fp(a["b_fc"]), _fp(a["W_out"]), _fp(a["b_out"])
        if "b_aa" in a:    # one-hot rows: onehot.W_aa + b_aa is the row W_aa[letter] + b_aa
            a["W_aa"] = np.ascontiguousarray(a["W_aa"] + a["b_aa"][None, :], dtype=np.float32)
            g.W_aa = _fp(a["W_aa"])
        g.embed_linear = 1 if topo["embed_linear"] else 0
        g.lm_dim = topo["lm_dim"]
        g.W_lm = _fp(a["W_lm"]) if g.lm_dim else NULL
        g.b_lm = _fp(a["b_lm"]) if g.lm_dim else NULL
        _check(mdf_model_create(&g, 0, &self._gcn))
        if g.lm_dim:
            l.hidden = a["lm_U1"].shape[0]
            l.W1, l.U1, l.b1 = _fp(a["lm_W1"]), _fp(a["lm_U1"]), _fp(a["lm_b1"])
            l.W2, l.U2, l.b2 = _fp(a["lm_W2"]), _fp(a["lm_U2"]), _fp(a["lm_b2"])
            _check(mdf_lm_create(&l, 0, &self._lm))
            _check(mdf_model_attach_lm(self._gcn, self._lm))
        self._T = mdf_model_num_terms(self._gcn)

    cdef _load_cnn(self, dict a, dict topo):
        cdef mdf_cnn_weights c
        cdef int n = len(topo["filters"]), k
        cdef int32_t *meta = <int32_t *>malloc(3 * n * sizeof(int32_t))
        cdef const float **ptrs = <const float **>malloc(2 * n * sizeof(const float *))
        if meta == NULL or ptrs == NULL:
            free(meta)
            free(ptrs)
            raise MemoryError()
        try:
            for k in range(n):
                meta[k] = topo["kernel_lens"][k]
                meta[n + k] = topo["filters"][k]
                meta[2 * n + k] = int(np.asarray(a.get(f"cnn_pad{k + 1}", (topo["kernel_lens"][k] - 1) // 2)).reshape(-1)[0])
                ptrs[k] = _fp(a[f"cnn_W{k + 1}"])
                ptrs[n + k] = _fp(a[f"cnn_b{k + 1}"])
            c.n_branch, c.kernel_len, c.filters, c.pad_left = n, meta, meta + n, meta + 2 * n
            c.W, c.b = ptrs, ptrs + n
            c.bn_gamma, c.bn_beta, c.bn_mean, c.bn_var = _fp(a["bn_gamma"]), _fp(a["bn_beta"]), _fp(a["bn_mean"]), _fp(a["bn_var"])
            c.bn_eps = float(np.asarray(a.get("bn_eps", 1e-3)).reshape(-1)[0])
            c.n_terms = topo["n_terms"]
            c.W_out, c.b_out = _fp(a["W_out"]), _fp(a["b_out"])
            _check(mdf_cnn_create(&c, 0, &self._cnn))
        finally:
            free(meta)
            free(ptrs)
        self._T = mdf_cnn_num_terms(self._cnn)

    def forward_pass(self, seqres: str, cmap = None):
        cdef bytes b = seqres.encode("ascii")
        cdef Py_ssize_t L = len(b)
        cdef cnp.ndarray[float, ndim=1, mode="c"] y = np.empty(self._T, dtype=np.float32)
        cdef cnp.ndarray A
        cdef int64_t bad = -1
        cdef int rc
        cdef const char *sp = b
        cdef float *yp = <float *>y.data
        cdef const void *ap
        if L == 0:
            raise ValueError("empty sequence")
        if cmap is not None:
            if self._gcn == NULL:
                raise ValueError("this is a sequence-only (CNN) model: it takes no contact map")
            A = np.ascontiguousarray(cmap, dtype=np.float32)        # the reference casts with astype(np.float32) at this spot
            if A.ndim != 2 or A.shape[0] != L or A.shape[1] != L:
                raise ValueError(f"cmap has shape {(<object>A).shape}, expected ({L}, {L})")
            ap = <const void *>A.data
            with nogil:
                rc = mdf_gcn_forward_host(self._gcn, sp, L, ap, MDF_DT_F32, yp, &bad)
        else:
            if self._cnn == NULL:
                raise ValueError("this is a GCN model (inputs cmap, seq): pass the contact map")
            with nogil:
                rc = mdf_cnn_forward_host(self._cnn, sp, L, yp, &bad)
        if rc == MDF_EBADCHAR:
            raise ValueError(f"Invalid character in sequence: {seqres[bad]}")
        _check(rc)
        return y

    def __dealloc__(self):
        if self._gcn != NULL:
            mdf_model_free(self._gcn)
        if self._lm != NULL:
            mdf_lm_free(self._lm)
        if self._cnn != NULL:
            mdf_cnn_free(self._cnn)


def predict_batch(list predictors, list seqs, list coords, list q_alns, list t_alns, double threshold = 6.0, int generated_contacts = 2,
                  int max_rows = 262144):   # (= MDF_DEFAULT_CHUNK_ROWS, include/mdfri.h)
    """The batched counterpart of the reference's two loops -- Pool.map(build_align_contact_map) (pipeline.py:476-481) followed by
    _run_prediction_loop (pipeline.py:292-319) -- as ONE call into the library (mdf_engine_run_alignments_host): C-alpha
    coordinates + gapped alignments + sequences of B proteins in, one (B, T) float32 array per GCN Predictor out.  The arrays
    cross the boundary as packed bytes / floats with per-protein lengths; planning, upload, the fused launch sequence,
    validation (with the CSR-capacity retry) and the download happen inside."""
    cdef int n = len(predictors), B = len(seqs), k
    cdef mdf_model *models[MAX_HEADS]
    cdef float *outs[MAX_HEADS]
    cdef mdf_engine *eng = NULL
    cdef mdf_engine_config cfg
    cdef int64_t info[4]
    cdef int rc
    cdef Predictor pr
    if n == 0 or n > MAX_HEADS:
        raise ValueError(f"1..{MAX_HEADS} predictors expected")
    if not (len(coords) == len(q_alns) == len(t_alns) == B) or B == 0:
        raise ValueError("seqs, coords, q_alns and t_alns must be non-empty lists of one length")
    for k in range(n):
        pr = predictors[k]
        if pr._gcn == NULL:
            raise ValueError("predict_batch takes GCN predictors")
        models[k] = pr._gcn
    cdef bytes sb = "".join(seqs).encode("ascii"), qb = "".join(q_alns).encode("ascii"), tb = "".join(t_alns).encode("ascii")
    cdef cnp.ndarray[int32_t, ndim=1, mode="c"] Lq = np.array([len(x) for x in seqs], dtype=np.int32)
    cdef cnp.ndarray[int32_t, ndim=1, mode="c"] La = np.array([len(x) for x in q_alns], dtype=np.int32)
    if [len(x) for x in t_alns] != La.tolist():
        raise ValueError("gapped query and target differ in length")
    cs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 3) for c in coords]
    cdef cnp.ndarray[int32_t, ndim=1, mode="c"] Lt = np.array([c.shape[0] for c in cs], dtype=np.int32)
    cdef cnp.ndarray[float, ndim=2, mode="c"] xyz = np.ascontiguousarray(np.concatenate(cs, axis=0)) if int(Lt.sum()) else np.zeros((1, 3), np.float32)
    results = [np.empty((B, (<Predictor>predictors[k])._T), dtype=np.float32) for k in range(n)]
    for k in range(n):
        outs[k] = <float *>cnp.PyArray_DATA(results[k])
    cfg.max_rows, cfg.nnz_per_row, cfg.threshold, cfg.generated_contacts = max_rows, 0, threshold, generated_contacts
    cfg.max_segment_groups, cfg.lm_batch, cfg.lm_workspace_gib, cfg.graph_max_chunks, cfg.pipeline_contact = 0, 0, 0.0, 0, 0
    _check(mdf_engine_create(models, n, 0, &cfg, &eng))
    cdef const char *sp = sb
    cdef const char *qp = qb
    cdef const char *tp = tb
    cdef const int32_t *lqp = <const int32_t *>Lq.data
    cdef const int32_t *ltp = <const int32_t *>Lt.data
    cdef const int32_t *lap = <const int32_t *>La.data
    cdef const float *xp = <const float *>xyz.data
    try:
        with nogil:
            rc = mdf_engine_run_alignments_host(eng, sp, lqp, B, xp, ltp, qp, tp, lap, outs, info)
        if rc == MDF_EBADCHAR:
            raise ValueError(f"Invalid character in sequence: {seqs[info[0]][info[1]]}")
        _check(rc)
    finally:
        mdf_engine_free(eng)
    return results


cdef class BatchEngine:
    """predict_batch for a caller with MANY batches: one engine for the object's life (predict_batch above makes and frees one per call)
    and the library's two-slot host pipeline (mdf_engine_submit_alignments_host / mdf_engine_collect_host): `submit` returns as soon as
    the batch is packed and enqueued, `collect` hands out the score arrays of the oldest batch in flight -- so the Python packing of batch
    k + 1 and the unpacking of batch k - 1 run under the kernels of batch k.  `run(batches)` is the loop; `predict_batch` is submit + collect.

        eng = BatchEngine([mf, bp, cc])
        for scores in eng.run(batches):      # batches: iterable of (seqs, coords, q_alns, t_alns); scores: one (B, T) float32 array per Predictor
            ...
    """
    cdef mdf_engine *_eng
    cdef int _n
    cdef list _predictors, _inflight

    def __cinit__(self):
        self._eng = NULL
        self._n = 0

    def __init__(self, list predictors, double threshold = 6.0, int generated_contacts = 2, int max_rows = 262144):   # (= MDF_DEFAULT_CHUNK_ROWS)
        cdef mdf_model *models[MAX_HEADS]
        cdef mdf_engine_config cfg
        cdef Predictor pr
        cdef int k, n = len(predictors)
        if n == 0 or n > MAX_HEADS:
            raise ValueError(f"1..{MAX_HEADS} predictors expected")
        for k in range(n):
            pr = predictors[k]
            if pr._gcn == NULL:
                raise ValueError("BatchEngine takes GCN predictors")
            models[k] = pr._gcn
        cfg.max_rows, cfg.nnz_per_row, cfg.threshold, cfg.generated_contacts = max_rows, 0, threshold, generated_contacts
        cfg.max_segment_groups, cfg.lm_batch, cfg.lm_workspace_gib, cfg.graph_max_chunks, cfg.pipeline_contact = 0, 0, 0.0, 0, 0
        _check(mdf_engine_create(models, n, 0, &cfg, &self._eng))
        self._n = n
        self._predictors = list(predictors)   # (the engine borrows their model handles)
        self._inflight = []

    def __dealloc__(self):
        if self._eng != NULL:
            mdf_engine_free(self._eng)

    def submit(self, list seqs, list coords, list q_alns, list t_alns):
        cdef int B = len(seqs), rc
        cdef int64_t ticket = -1
        if not (len(coords) == len(q_alns) == len(t_alns) == B) or B == 0:
            raise ValueError("seqs, coords, q_alns and t_alns must be non-empty lists of one length")
        cdef bytes sb = "".join(seqs).encode("ascii"), qb = "".join(q_alns).encode("ascii"), tb = "".join(t_alns).encode("ascii")
        cdef cnp.ndarray[int32_t, ndim=1, mode="c"] Lq = np.fromiter(map(len, seqs), dtype=np.int32, count=B)
        cdef cnp.ndarray[int32_t, ndim=1, mode="c"] La = np.fromiter(map(len, q_alns), dtype=np.int32, count=B)
        if not np.array_equal(La, np.fromiter(map(len, t_alns), dtype=np.int32, count=B)):
            raise ValueError("gapped query and target differ in length")
        cs = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1, 3) for c in coords]
        cdef cnp.ndarray[int32_t, ndim=1, mode="c"] Lt = np.fromiter((c.shape[0] for c in cs), dtype=np.int32, count=B)
        cdef cnp.ndarray[float, ndim=2, mode="c"] xyz = np.ascontiguousarray(np.concatenate(cs, axis=0)) if int(Lt.sum()) else np.zeros((1, 3), np.float32)
        cdef const char *sp = sb
        cdef const char *qp = qb
        cdef const char *tp = tb
        cdef const int32_t *lqp = <const int32_t *>Lq.data
        cdef const int32_t *ltp = <const int32_t *>Lt.data
        cdef const int32_t *lap = <const int32_t *>La.data
        cdef const float *xp = <const float *>xyz.data
        with nogil:
            rc = mdf_engine_submit_alignments_host(self._eng, sp, lqp, B, xp, ltp, qp, tp, lap, &ticket)
        _check(rc)
        self._inflight.append((ticket, B, seqs))
        return ticket

    def collect(self):
        """Score arrays of the oldest batch in flight: one (B, T) float32 array per Predictor, rows in the order the lists were given."""
        cdef float *outs[MAX_HEADS]
        cdef int64_t info[4]
        cdef int64_t ticket
        cdef int k, rc, B
        if not self._inflight:
            raise ValueError("no batch in flight")
        t, B, seqs = self._inflight.pop(0)
        ticket = t
        results = [np.empty((B, (<Predictor>self._predictors[k])._T), dtype=np.float32) for k in range(self._n)]
        for k in range(self._n):
            outs[k] = <float *>cnp.PyArray_DATA(results[k])
        with nogil:
            rc = mdf_engine_collect_host(self._eng, ticket, outs, info)
        if rc == MDF_EBADCHAR:
            raise ValueError(f"Invalid character in sequence: {seqs[info[0]][info[1]]}")
        _check(rc)
        return results

    def predict_batch(self, list seqs, list coords, list q_alns, list t_alns):
        self.submit(seqs, coords, q_alns, t_alns)
        return self.collect()

    def run(self, batches):
        for item in batches:
            self.submit(*item)
            if len(self._inflight) == 2:
                yield self.collect()
        while self._inflight:
            yield self.collect()
