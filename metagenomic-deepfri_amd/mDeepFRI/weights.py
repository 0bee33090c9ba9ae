"""Weight containers for the DeepFRI GCN (the arithmetic the reference keeps in external .onnx files,
reference mDeepFRI/__init__.py:47-80, mDeepFRI/utils.py:119-151).

`.mdfw` is the native container read by libmdfri_hip (mdf_model_load):
    bytes 0..7   b"MDFW0001"
    u32          n_tensors
    n x entry    { char name[32] (NUL padded); u32 ndim; u64 dims[4]; u64 offset_from_file_start }
    raw little-endian float32 data (each tensor 64-byte aligned)
Tensor names: W_aa (26,E)  W_gc1 (E,C1)  W_gc2 (C1,C2)  W_gc3 (C2,C3)  W_fc (C1+C2+C3,F)  b_fc (F)
              W_out (F,2T)  b_out (2T)        -- Keras orientation (in_features, out_features).
Optional language-model branch (the released models have it), Keras orientation and gate
order i,f,c,o:  W_lm (H,E)  b_lm (E)  lm_W1 (26,4H)  lm_U1 (H,4H)  lm_b1 (4H)  lm_W2 (H,4H)  lm_U2 (H,4H)  lm_b2 (4H).
Sequence-only CNN models (`DeepCNN-MERGED_{mode}`, reference __init__.py:68; run when cmap is None, predict.pyx:91-95) use
another key set: cnn_W{b} (k_b,26,F_b)  cnn_b{b} (F_b)  [cnn_pad{b} scalar]  bn_gamma/bn_beta/bn_mean/bn_var (sum F_b)
bn_eps (1)  W_out (sum F_b, 2T)  b_out (2T);  `model_kind(weights)` tells the two apart.
`.npz` files with the same keys are accepted too.

`.onnx` files (what the reference's pipeline passes, pipeline.py:549-584) are read by mDeepFRI.onnx_reader -- a
structural extraction of the tensors above from the tf2onnx graph, no onnx / onnxruntime needed.  `resolve_model_path`
prefers a `.mdfw`/`.npz` sibling of the given path (convert once with `python -m mDeepFRI.onnx_reader model.onnx`) and
falls back to parsing the `.onnx` itself.  The reader has been exercised on schema-conformant files only, never on a
released DeepFRI file (none is available offline): it refuses graphs it does not recognise.
"""
from __future__ import annotations

import os
import struct

import numpy as np

MAGIC = b"MDFW0001"
ORDER = ("W_aa", "W_gc1", "W_gc2", "W_gc3", "W_fc", "b_fc", "W_out", "b_out",
         "W_lm", "b_lm", "lm_W1", "lm_U1", "lm_b1", "lm_W2", "lm_U2", "lm_b2")
LM_KEYS = ("lm_W1", "lm_U1", "lm_b1", "lm_W2", "lm_U2", "lm_b2")
_ENTRY = struct.Struct("<32sI4QQ")


def save_mdfw(path: str, weights: dict) -> None:
    names = [k for k in ORDER if k in weights] + sorted(k for k in weights if k not in ORDER)
    arrays = [np.ascontiguousarray(weights[k], dtype="<f4") for k in names]
    head = len(MAGIC) + 4 + _ENTRY.size * len(names)
    offsets, off = [], (head + 63) // 64 * 64
    for a in arrays:
        offsets.append(off)
        off = (off + a.nbytes + 63) // 64 * 64
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(names)))
        for n, a, o in zip(names, arrays, offsets):
            dims = list(a.shape) + [0] * (4 - a.ndim)
            f.write(_ENTRY.pack(n.encode(), a.ndim, *dims, o))
        for a, o in zip(arrays, offsets):
            f.seek(o)
            f.write(a.tobytes())


def mdfw_names(path: str) -> list:
    """Tensor names of a .mdfw container (directory only)."""
    with open(path, "rb") as f:
        head = f.read(12)
        if head[:8] != MAGIC:
            raise OSError(f"{path} is not an MDFW0001 container")
        (n,) = struct.unpack_from("<I", head, 8)
        d = f.read(n * _ENTRY.size)
    return [_ENTRY.unpack_from(d, i * _ENTRY.size)[0].rstrip(b"\0").decode() for i in range(n)]


def load_mdfw(path: str) -> dict:
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != MAGIC:
        raise OSError(f"{path} is not an MDFW0001 container")
    (n,) = struct.unpack_from("<I", buf, 8)
    out = {}
    for i in range(n):
        name, ndim, d0, d1, d2, d3, off = _ENTRY.unpack_from(buf, 12 + i * _ENTRY.size)
        shape = (d0, d1, d2, d3)[:ndim]
        cnt = int(np.prod(shape))
        out[name.rstrip(b"\0").decode()] = np.frombuffer(buf, dtype="<f4", count=cnt, offset=off).reshape(shape).copy()
    return out


def model_kind(weights: dict) -> str:
    """"cnn" for a sequence-only DeepCNN weight set, "gcn" otherwise."""
    return "cnn" if "cnn_W1" in weights else "gcn"


def validate_cnn(weights: dict) -> dict:
    """Check a CNN weight set and return {kind, kernel_lens, filters, channels, n_terms}."""
    w = weights
    ks, fs, b = [], [], 1
    while f"cnn_W{b}" in w:
        W = w[f"cnn_W{b}"]
        if W.ndim != 3 or W.shape[1] != 26:
            raise ValueError(f"weights: cnn_W{b} must be (kernel_len, 26, filters), got {tuple(W.shape)}")
        if f"cnn_b{b}" not in w or tuple(w[f"cnn_b{b}"].shape) != (W.shape[2],):
            raise ValueError(f"weights: cnn_b{b} missing or not ({W.shape[2]},)")
        ks.append(int(W.shape[0]))
        fs.append(int(W.shape[2]))
        b += 1
    C = sum(fs)
    for k in ("bn_gamma", "bn_beta", "bn_mean", "bn_var"):
        if k not in w or tuple(w[k].shape) != (C,):
            raise ValueError(f"weights: {k} missing or not ({C},)")
    if "W_out" not in w or "b_out" not in w or w["W_out"].ndim != 2 or w["W_out"].shape[0] != C or w["W_out"].shape[1] % 2 \
            or tuple(w["b_out"].shape) != (w["W_out"].shape[1],):
        raise ValueError("weights: W_out / b_out shape mismatch")
    return {"kind": "cnn", "kernel_lens": ks, "filters": fs, "channels": C, "n_terms": int(w["W_out"].shape[1] // 2)}


def validate(weights: dict) -> dict:
    """Check shapes and return the topology {embed, gc_dims, fc_dim, n_terms, lm_dim} (GCN) -- or validate_cnn's for a CNN."""
    w = weights
    if model_kind(w) == "cnn":
        return validate_cnn(w)
    for k in ("W_aa", "W_gc1", "W_fc", "b_fc", "W_out", "b_out"):
        if k not in w:
            raise ValueError(f"weights: missing tensor {k}")
    if w["W_aa"].ndim != 2 or w["W_aa"].shape[0] != 26:
        raise ValueError("weights: W_aa must be (26, E)")
    prev, gc = w["W_aa"].shape[1], []
    k = 1
    while f"W_gc{k}" in w:
        a = w[f"W_gc{k}"]
        if a.ndim != 2 or a.shape[0] != prev:
            raise ValueError(f"weights: W_gc{k} has shape {a.shape}, expected ({prev}, C)")
        prev = a.shape[1]
        gc.append(prev)
        k += 1
    if not 1 <= len(gc) <= 3:
        raise ValueError("weights: 1..3 GraphConv layers supported")
    if w["W_fc"].shape != (sum(gc), w["b_fc"].shape[0]):
        raise ValueError("weights: W_fc / b_fc shape mismatch")
    if w["W_out"].shape[0] != w["W_fc"].shape[1] or w["W_out"].shape[1] % 2 or w["b_out"].shape[0] != w["W_out"].shape[1]:
        raise ValueError("weights: W_out / b_out shape mismatch")
    lm_dim = 0
    if "W_lm" in w or any(k in w for k in LM_KEYS):
        for k in ("W_lm", "b_lm") + LM_KEYS:
            if k not in w:
                raise ValueError(f"weights: language-model branch is incomplete, missing {k}")
        lm_dim = int(w["W_lm"].shape[0])
        E = int(w["W_aa"].shape[1])
        if w["W_lm"].shape != (lm_dim, E) or w["b_lm"].shape != (E,):
            raise ValueError("weights: W_lm / b_lm shape mismatch")
        H4 = 4 * lm_dim
        want = {"lm_W1": (26, H4), "lm_U1": (lm_dim, H4), "lm_b1": (H4,), "lm_W2": (lm_dim, H4), "lm_U2": (lm_dim, H4), "lm_b2": (H4,)}
        for k, shp in want.items():
            if tuple(w[k].shape) != shp:
                raise ValueError(f"weights: {k} has shape {tuple(w[k].shape)}, expected {shp}")
    if "b_aa" in w and tuple(w["b_aa"].shape) != (int(w["W_aa"].shape[1]),):
        raise ValueError("weights: b_aa must be (E,)")
    # optional 1-element tensor: the embedding has NO activation (a topology detail the model file decides, see include/mdfri.h)
    embed_linear = bool("embed_linear" in w and float(np.asarray(w["embed_linear"]).reshape(-1)[0]) != 0.0)
    return {"embed": int(w["W_aa"].shape[1]), "gc_dims": gc, "fc_dim": int(w["W_fc"].shape[1]),
            "n_terms": int(w["W_out"].shape[1] // 2), "lm_dim": lm_dim, "embed_linear": embed_linear}


def resolve_model_path(model_path: str) -> str:
    """Map the path the pipeline passes to Predictor (an .onnx file name, reference pipeline.py:549-584) to a
    container this build can read."""
    if model_path.endswith((".mdfw", ".npz")):
        if not os.path.exists(model_path):
            raise FileNotFoundError(model_path)
        return model_path
    stem = os.path.splitext(model_path)[0]
    for ext in (".mdfw", ".npz"):
        if os.path.exists(stem + ext):
            return stem + ext
    if model_path.endswith(".onnx") and os.path.exists(model_path):
        return model_path
    raise FileNotFoundError(
        f"{model_path!r}: no such model file, and no '.mdfw' / '.npz' weight container with the same stem")


def load_weights(model_path: str) -> dict:
    p = resolve_model_path(model_path)
    if p.endswith(".npz"):
        with np.load(p) as z:
            w = {k: np.asarray(z[k], dtype=np.float32) for k in z.files}
    elif p.endswith(".onnx"):
        from . import onnx_reader
        w = onnx_reader.extract_weights(onnx_reader.parse_model(p))
    else:
        w = load_mdfw(p)
    validate(w)
    return w
