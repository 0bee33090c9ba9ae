"""CPU restatement of the LSTM language-model branch of the released DeepFRI GCN models.

TEST INFRASTRUCTURE -- the checker for the HIP LSTM / embedding kernels, never the product.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

*** PARITY UNPINNED ***  Like the rest of the GCN arithmetic (see gcn_oracle.py) this branch is not in
/root/reference: it lives inside the external `.onnx` files the reference downloads (mDeepFRI/__init__.py:47-80) and
runs through onnxruntime (mDeepFRI/predict.pyx:63-73,98).  The only trace of it in the reference tree is the model
diagram docs/_images/deepfri_overview.png and the timing table weight_convert/inference_times.csv.gz.  What is restated
is the published architecture of flatironinstitute/DeepFRI (deepfrier/DeepFRI.py, Gligorijevic et al. 2021, Methods):

    lm   = Keras model  Input(None,26) -> LSTM(512, return_sequences, name="LSTM1") -> LSTM(512, ..., name="LSTM2")
           (frozen; the GCN takes the LSTM2 output sequence)
    x_lm = Dense(1024, use_bias=True,  name="LM_embedding")(lm(seq))
    x_aa = Dense(1024, use_bias=False, name="AA_embedding")(seq)
    X0   = relu(x_lm + x_aa)              -> GraphConv stack exactly as gcn_oracle.gcn_forward

Keras LSTM cell (tf.keras defaults: activation tanh, recurrent_activation sigmoid, unit_forget_bias only affects
initialisation), kernel W (I,4H), recurrent kernel U (H,4H), bias b (4H), gate blocks in the order i, f, c, o:
    z = x_t W + h_{t-1} U + b ;  i = sigmoid(z_i)  f = sigmoid(z_f)  g = tanh(z_c)  o = sigmoid(z_o)
    c_t = f * c_{t-1} + i * g ;  h_t = o * tanh(c_t) ;  h_0 = c_0 = 0
(ONNX's LSTM operator orders the blocks i, o, f, c: a converter from a real file must permute, see
mDeepFRI/onnx_reader.py.)

Weight keys: lm_W1 (26,4H) lm_U1 (H,4H) lm_b1 (4H) lm_W2 (H,4H) lm_U2 (H,4H) lm_b2 (4H)  W_lm (H,E)  b_lm (E).
"""
import numpy as np

import gcn_oracle

LM_KEYS = ("lm_W1", "lm_U1", "lm_b1", "lm_W2", "lm_U2", "lm_b2")


def _sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def lstm_forward(x: np.ndarray, W: np.ndarray, U: np.ndarray, b: np.ndarray) -> np.ndarray:
    """x (L, I) -> h (L, H), one sequence, zero initial state."""
    dt = x.dtype
    H = U.shape[0]
    zx = x @ W + b  # input projection of every step at once
    h = np.zeros(H, dtype=dt)
    c = np.zeros(H, dtype=dt)
    out = np.empty((x.shape[0], H), dtype=dt)
    for t in range(x.shape[0]):
        z = zx[t] + h @ U
        i, f = _sigmoid(z[:H]), _sigmoid(z[H:2 * H])
        g, o = np.tanh(z[2 * H:3 * H]), _sigmoid(z[3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        out[t] = h
    return out


def lm_forward(weights: dict, seq: str, dtype=np.float32) -> np.ndarray:
    """LSTM2 output sequence (L, H) for one protein."""
    dt = np.dtype(dtype)
    w = {k: np.asarray(weights[k], dtype=dt) for k in LM_KEYS}
    S = gcn_oracle.onehot(seq, dt)
    h1 = lstm_forward(S, w["lm_W1"], w["lm_U1"], w["lm_b1"])
    return lstm_forward(h1, w["lm_W2"], w["lm_U2"], w["lm_b2"])


def gcn_lm_forward(weights: dict, seq: str, cmap: np.ndarray, dtype=np.float32, return_intermediates=False):
    """Scores of one protein through the GCN with the language-model branch."""
    dt = np.dtype(dtype)
    w = {k: np.asarray(v, dtype=dt) for k, v in weights.items()}
    S = gcn_oracle.onehot(seq, dt)
    h2 = lm_forward(weights, seq, dt)
    A = np.asarray(cmap).reshape(len(seq), len(seq)).astype(np.float32).astype(dt)
    X = S @ w["W_aa"] + (h2 @ w["W_lm"] + w["b_lm"])
    if "b_aa" in w:
        X = X + w["b_aa"]
    if not ("embed_linear" in w and float(np.asarray(w["embed_linear"]).reshape(-1)[0]) != 0.0):
        X = np.maximum(X, 0)
    A_hat = gcn_oracle.normalize_adjacency(A)
    feats = []
    k = 1
    while f"W_gc{k}" in w:
        X = gcn_oracle.elu((A_hat @ X) @ w[f"W_gc{k}"])
        feats.append(X)
        k += 1
    g = np.concatenate(feats, axis=1).sum(axis=0, dtype=dt)
    f = np.maximum(g @ w["W_fc"] + w["b_fc"], 0)
    z = (f @ w["W_out"] + w["b_out"]).reshape(-1, 2)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    y = (e / e.sum(axis=1, keepdims=True))[:, 0].astype(dt)
    if return_intermediates:
        return y, {"h2": h2, "H": feats, "g": g}
    return y
