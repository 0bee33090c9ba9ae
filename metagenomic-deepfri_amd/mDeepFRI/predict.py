"""Drop-in for the reference's Cython module mDeepFRI/predict.pyx: `seq2onehot` and `Predictor`, with the ONNX
Runtime session replaced by hand-written HIP kernels (libmdfri_hip.so).  No onnxruntime import."""
from __future__ import annotations

import ctypes
import hashlib
import os
import weakref

import numpy as np

from . import _hip, weights as _weights

__all__ = ["seq2onehot", "Predictor"]

_DTYPES = {np.dtype(np.int32): _hip.DT_I32, np.dtype(np.float32): _hip.DT_F32, np.dtype(np.int64): _hip.DT_I64,
           np.dtype(np.float64): _hip.DT_F64, np.dtype(np.uint8): _hip.DT_U8}


def seq2onehot(seq: str) -> np.ndarray:
    """reference predict.pyx:17-48: float32 (L,26) one-hot in the order "-DGULNTKHYWCPVSOIEFXQABZRM";
    ValueError(f"Invalid character in sequence: {c}") for any other byte; "" -> (0,26)."""
    if not isinstance(seq, str):
        raise TypeError(f"Argument 'seq' has incorrect type (expected str, got {type(seq).__name__})")
    b = seq.encode("ascii")
    out = np.zeros((len(b), 26), dtype=np.float32)
    if len(b) == 0:
        return out
    bad = _hip.c_int64(-1)
    rc = _hip.lib().mdf_seq2onehot(b, len(b), _hip.ptr(out), bad)
    if rc == _hip.MDF_EBADCHAR:
        raise ValueError(f"Invalid character in sequence: {seq[bad.value]}")
    _hip.check(rc)
    return out


class _Session:
    """Owns the mdf_model handle (the role onnxruntime.InferenceSession plays in predict.pyx:63-73)."""

    class _Input:
        def __init__(self, name):
            self.name = name

    def __init__(self, handle, topology, lm=None, kind="gcn"):
        self.handle = handle
        self.topology = topology
        self.lm = lm  # LanguageModel the handle points at (kept alive here), or None
        self.kind = kind  # "gcn": mdf_model (inputs cmap, seq);  "cnn": mdf_cnn (input seq only, predict.pyx:91-95)

    def get_inputs(self):
        return [self._Input("seq")] if self.kind == "cnn" else [self._Input("cmap"), self._Input("seq")]

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                (_hip.lib().mdf_cnn_free if self.kind == "cnn" else _hip.lib().mdf_model_free)(h)
            except Exception:
                pass


class LanguageModel:
    """The frozen LSTM language model inside the released DeepFRI GCN files (two stacked LSTM layers; part of the ONNX
    graph the reference runs at predict.pyx:98).  Every GO head's file carries the same copy, so identical weights
    share one device-side mdf_lm (`LanguageModel.shared`)."""

    _shared = weakref.WeakValueDictionary()

    def __init__(self, weights: dict, device: int = 0):
        w = {k: np.ascontiguousarray(weights[k], dtype=np.float32) for k in _weights.LM_KEYS}
        self.hidden = int(w["lm_U1"].shape[0])
        self.device = int(device)
        s = _hip.LmWeights()
        s.hidden = self.hidden
        fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))  # noqa: E731
        s.W1, s.U1, s.b1 = fp(w["lm_W1"]), fp(w["lm_U1"]), fp(w["lm_b1"])
        s.W2, s.U2, s.b2 = fp(w["lm_W2"]), fp(w["lm_U2"]), fp(w["lm_b2"])
        self.handle = ctypes.c_void_p()
        _hip.check(_hip.lib().mdf_lm_create(ctypes.byref(s), self.device, ctypes.byref(self.handle)))

    @classmethod
    def shared(cls, weights: dict, device: int = 0) -> "LanguageModel":
        h = hashlib.sha1()
        for k in _weights.LM_KEYS:
            h.update(np.ascontiguousarray(weights[k], dtype=np.float32).tobytes())
        key = (int(device), h.hexdigest())
        lm = cls._shared.get(key)
        if lm is None:
            lm = cls(weights, device)
            cls._shared[key] = lm
        return lm

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                _hip.lib().mdf_lm_free(h)
            except Exception:
                pass


def create_model_handle(weights: dict, device: int = 0):
    """weights dict (see mDeepFRI.weights) -> (mdf_model* as c_void_p, topology)."""
    topo = _weights.validate(weights)
    w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
    if "b_aa" in w:   # a bias on AA_embedding: the input rows are one-hot, so onehot.W_aa + b_aa IS the row W_aa[a] + b_aa (one rounding, as in the graph)
        w["W_aa"] = np.ascontiguousarray(w["W_aa"] + w["b_aa"][None, :], dtype=np.float32)
    s = _hip.GcnWeights()
    s.embed_linear = int(topo["embed_linear"])
    s.embed, s.n_gc, s.fc_dim, s.n_terms = topo["embed"], len(topo["gc_dims"]), topo["fc_dim"], topo["n_terms"]
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))  # noqa: E731
    for k, c in enumerate(topo["gc_dims"]):
        s.gc_dims[k] = c
        s.W_gc[k] = fp(w[f"W_gc{k + 1}"])
    s.W_aa, s.W_fc, s.b_fc, s.W_out, s.b_out = fp(w["W_aa"]), fp(w["W_fc"]), fp(w["b_fc"]), fp(w["W_out"]), fp(w["b_out"])
    s.lm_dim = topo["lm_dim"]
    if topo["lm_dim"]:
        s.W_lm, s.b_lm = fp(w["W_lm"]), fp(w["b_lm"])
    handle = ctypes.c_void_p()
    _hip.check(_hip.lib().mdf_model_create(ctypes.byref(s), int(device), ctypes.byref(handle)))
    return handle, topo


def create_cnn_handle(weights: dict, device: int = 0):
    """CNN weight dict (see mDeepFRI.weights) -> (mdf_cnn* as c_void_p, topology)."""
    topo = _weights.validate_cnn(weights)
    n = len(topo["filters"])
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))  # noqa: E731
    keep = []

    def f32(key):
        a = np.ascontiguousarray(weights[key], dtype=np.float32)
        keep.append(a)
        return a

    s = _hip.CnnWeights()
    s.n_branch = n
    klen = (ctypes.c_int32 * n)(*topo["kernel_lens"])
    filt = (ctypes.c_int32 * n)(*topo["filters"])
    pads = (ctypes.c_int32 * n)(*[int(np.asarray(weights.get(f"cnn_pad{b + 1}", (topo["kernel_lens"][b] - 1) // 2)).reshape(-1)[0])
                                  for b in range(n)])
    Ws = (ctypes.POINTER(ctypes.c_float) * n)(*[fp(f32(f"cnn_W{b + 1}")) for b in range(n)])
    bs = (ctypes.POINTER(ctypes.c_float) * n)(*[fp(f32(f"cnn_b{b + 1}")) for b in range(n)])
    s.kernel_len, s.filters, s.pad_left, s.W, s.b = klen, filt, pads, Ws, bs
    s.bn_gamma, s.bn_beta, s.bn_mean, s.bn_var = fp(f32("bn_gamma")), fp(f32("bn_beta")), fp(f32("bn_mean")), fp(f32("bn_var"))
    s.bn_eps = float(np.asarray(weights.get("bn_eps", 1e-3)).reshape(-1)[0])
    s.n_terms = topo["n_terms"]
    s.W_out, s.b_out = fp(f32("W_out")), fp(f32("b_out"))
    handle = ctypes.c_void_p()
    _hip.check(_hip.lib().mdf_cnn_create(ctypes.byref(s), int(device), ctypes.byref(handle)))
    return handle, topo


class Predictor(object):
    """reference predict.pyx:50-102.  Public attributes as there: model_path, threads, session, input_names.

    `model_path` is the path the pipeline hands over (reference pipeline.py:584); see
    mDeepFRI.weights.resolve_model_path for the containers accepted.  `threads` is kept for signature
    compatibility (the reference forwards it to ORT's CPU thread pools; there are none here).
    Extra keyword `device`: HIP device ordinal (default: $MDFRI_DEVICE or 0).  `weights=` builds a predictor from
    an in-memory weight dict (used with synthetic weights)."""

    def __init__(self, model_path: str, threads: int = 1, device: int | None = None, weights: dict | None = None):
        self.model_path = model_path
        self.threads = threads
        self.device = int(os.environ.get("MDFRI_DEVICE", "0")) if device is None else int(device)
        self._weights = weights
        self.session = None
        self.input_names = []
        self._load_model()

    def _load_model(self):
        L = _hip.lib()
        w = self._weights
        if w is None:
            path = _weights.resolve_model_path(self.model_path)
            if not (path.endswith(".mdfw") and "cnn_W1" not in _weights.mdfw_names(path)):
                w = _weights.load_weights(path)
        if w is not None and _weights.model_kind(w) == "cnn":   # sequence-only model (reference pipeline.py:600-648)
            handle, topo = create_cnn_handle(w, self.device)
            self.session = _Session(handle, {"n_terms": topo["n_terms"], "feature_dim": topo["channels"], "lm_dim": 0}, None, "cnn")
            self.input_names = [node.name for node in self.session.get_inputs()]
            self._weights = None
            return
        if w is not None:
            handle, _ = create_model_handle(w, self.device)
        else:  # native GCN container: read by the library itself
            handle = ctypes.c_void_p()
            _hip.check(L.mdf_model_load(path.encode(), self.device, ctypes.byref(handle)))
            if L.mdf_model_lm_dim(handle) > 0:
                w = _weights.load_mdfw(path)  # the language-model tensors are uploaded from here
        topo = {"n_terms": int(L.mdf_model_num_terms(handle)), "feature_dim": int(L.mdf_model_feature_dim(handle)),
                "lm_dim": int(L.mdf_model_lm_dim(handle))}
        lm = None
        if topo["lm_dim"]:
            lm = LanguageModel.shared(w, self.device)
            _hip.check(L.mdf_model_attach_lm(handle, lm.handle))
        self.session = _Session(handle, topo, lm)
        self.input_names = [node.name for node in self.session.get_inputs()]
        self._weights = None

    @property
    def n_terms(self) -> int:
        return self.session.topology["n_terms"]

    def forward_pass(self, seqres: str, cmap=None) -> np.ndarray:
        """reference predict.pyx:75-102, GCN branch: float32 (T,) = softmax(...)[:, :, 0].reshape(-1)."""
        if not isinstance(seqres, str):
            raise TypeError("seqres must be str")
        b = seqres.encode("ascii")
        L = len(b)
        if self.session.kind == "cnn":
            # reference predict.pyx:91-95: the one-hot sequence is the only input of the DeepCNN session
            if cmap is not None:
                raise ValueError("this is a sequence-only (CNN) model: it takes no contact map")
            if L == 0:
                raise ValueError("empty sequence")
            scores = np.empty((self.n_terms,), dtype=np.float32)
            bad = _hip.c_int64(-1)
            rc = _hip.lib().mdf_cnn_forward_host(self.session.handle, b, L, _hip.ptr(scores), bad)
            if rc == _hip.MDF_EBADCHAR:
                raise ValueError(f"Invalid character in sequence: {seqres[bad.value]}")
            _hip.check(rc)
            return scores
        if cmap is None:
            raise ValueError("this is a GCN model (inputs cmap, seq): pass the contact map; sequence-only prediction needs a "
                             "DeepCNN model file")
        A = np.asarray(cmap)
        if A.ndim != 2 or A.shape != (L, L):
            raise ValueError(f"cmap has shape {A.shape}, expected ({L}, {L}) for a sequence of length {L}")
        if L == 0:
            raise ValueError("empty sequence")
        if A.dtype == np.bool_:
            A = A.view(np.uint8)
        if A.dtype not in _DTYPES:
            A = A.astype(np.float32)  # predict.pyx:88 casts to float32 anyway
        A = np.ascontiguousarray(A)
        scores = np.empty((self.n_terms,), dtype=np.float32)
        bad = _hip.c_int64(-1)
        rc = _hip.lib().mdf_gcn_forward_host(self.session.handle, b, L, _hip.ptr(A), _DTYPES[A.dtype], _hip.ptr(scores), bad)
        if rc == _hip.MDF_EBADCHAR:
            raise ValueError(f"Invalid character in sequence: {seqres[bad.value]}")
        _hip.check(rc)
        return scores
