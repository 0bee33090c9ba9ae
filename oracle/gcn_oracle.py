"""CPU restatement of the DeepFRI GCN forward the reference runs through ONNX Runtime.

TEST INFRASTRUCTURE -- the checker for the HIP GCN kernels, never the product.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

*** PARITY UNPINNED ***  The arithmetic of this stage is not in /root/reference: it lives in external
`.onnx` files (URLs at reference mDeepFRI/__init__.py:47-80) executed by the un-vendored, unpinned
`onnxruntime` dependency (reference pyproject.toml:36-37; call sites mDeepFRI/predict.pyx:63-73,98).
Neither is present in this image and no reference test constructs a Predictor or checks a score
(mDeepFRI/tests/test_predict.py covers seq2onehot only).  What is restated here is therefore the
published algorithm of the model those files were converted from -- flatironinstitute/DeepFRI,
`deepfrier/DeepFRI.py` + `deepfrier/layers.py` (GraphConv / SumPooling / FuncPredictor), Keras graph
converted by tf2onnx opset 15 (reference weight_convert/convert_models2onnx.py:39-43) -- anchored on
the reference's own call site:
    inputs  {in0: cmap (1,L,L) f32, in1: onehot (1,L,26) f32}     predict.pyx:82-90
    output  prediction[:, :, 0].reshape(-1)  of a (1,T,2) tensor    predict.pyx:98-100
and on the model file name `GraphConv_gcd_512-512-512_fcd_1024` (mDeepFRI/__init__.py:73,78).
Scope follows BASELINE.json north_star: one-hot embed -> 3x GraphConv -> sum pool -> dense -> GO head
(the language-model branch of the released weights is a listed "next" row).

Op order (fp32 unless dtype=float64 is requested for tolerance studies):
    X0   = relu(S @ W_aa)                                   Dense(use_bias=False) + Activation('relu')
           (topology variants a released file may turn out to have, carried as DATA: `embed_linear` -- no activation;
            `b_aa` -- a bias on AA_embedding; mDeepFRI.onnx_reader sets them from the graph)
    A'   = A - diag(diag(A)) + I ;  d = 1/(1e-6 + sqrt(rowsum(A')))
    Ahat = (diag(d) @ A') @ diag(d)                         GraphConv._normalize
    H_k  = elu((Ahat @ H_{k-1}) @ W_k)      k=1..3          batch_dot then dot, use_bias=False, elu
    g    = sum_rows(concat(H_1,H_2,H_3))                    SumPooling(axis=1)
    f    = relu(g @ W_fc + b_fc)                            Dense(relu); Dropout = identity at inference
    z    = f @ W_out + b_out ; reshape (T,2) ; softmax(-1)  FuncPredictor
    y    = softmax[:, 0]                                    predict.pyx:100
"""
import numpy as np

ALPHABET = "-DGULNTKHYWCPVSOIEFXQABZRM"  # reference mDeepFRI/predict.pyx:26


def seq_to_index(seq: str) -> np.ndarray:
    lut = {c: i for i, c in enumerate(ALPHABET)}
    try:
        return np.array([lut[c] for c in seq], dtype=np.int32)
    except KeyError as e:  # mirrors predict.pyx:45-46
        raise ValueError(f"Invalid character in sequence: {e.args[0]}")


def onehot(seq: str, dtype=np.float32) -> np.ndarray:
    idx = seq_to_index(seq)
    S = np.zeros((len(seq), 26), dtype=dtype)
    S[np.arange(len(seq)), idx] = 1
    return S


def normalize_adjacency(A: np.ndarray, eps: float = 1e-6) -> np.ndarray:
    dt = A.dtype
    A = A - np.diag(np.diag(A))
    A_hat = A + np.eye(A.shape[0], dtype=dt)
    d = (dt.type(1.0) / (dt.type(eps) + np.sqrt(A_hat.sum(axis=1, dtype=dt)))).astype(dt)
    return (d[:, None] * A_hat) * d[None, :]


def elu(x):
    return np.where(x > 0, x, np.exp(np.minimum(x, 0)) - x.dtype.type(1.0)).astype(x.dtype)


def gcn_forward(weights: dict, seq: str, cmap: np.ndarray, dtype=np.float32, return_intermediates=False):
    """weights: W_aa (26,E), W_gc1 (E,C1), W_gc2 (C1,C2), W_gc3 (C2,C3), W_fc (C1+C2+C3,F), b_fc (F,),
    W_out (F,2T), b_out (2T,).  cmap: (L,L) any numeric dtype (cast to f32 first, predict.pyx:88)."""
    dt = np.dtype(dtype)
    w = {k: np.asarray(v, dtype=dt) for k, v in weights.items() if not isinstance(v, str)}
    S = onehot(seq, dt)
    A = np.asarray(cmap).reshape(len(seq), len(seq)).astype(np.float32).astype(dt)
    X = S @ w["W_aa"]
    if "b_aa" in w:                                                        # topology variant: AA_embedding with a bias
        X = X + w["b_aa"]
    if not ("embed_linear" in w and float(np.asarray(w["embed_linear"]).reshape(-1)[0]) != 0.0):
        X = np.maximum(X, 0)                                               # Activation('relu'); absent in the `embed_linear` variant
    A_hat = normalize_adjacency(A)
    feats = []
    k = 1
    while f"W_gc{k}" in w:
        X = elu((A_hat @ X) @ w[f"W_gc{k}"])
        feats.append(X)
        k += 1
    g = np.concatenate(feats, axis=1).sum(axis=0, dtype=dt)
    f = np.maximum(g @ w["W_fc"] + w["b_fc"], 0)
    z = (f @ w["W_out"] + w["b_out"]).reshape(-1, 2)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    p = e / e.sum(axis=1, keepdims=True)
    y = p[:, 0].astype(dt)
    if return_intermediates:
        return y, {"A_hat": A_hat, "H": feats, "g": g, "f": f, "z": z}
    return y
