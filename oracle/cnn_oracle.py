"""CPU restatement of the sequence-only DeepFRI CNN the reference runs when no contact map exists
(`Predictor.forward_pass(seqres)` with `cmap=None`, reference mDeepFRI/predict.pyx:91-100; caller pipeline.py:600-648).

TEST INFRASTRUCTURE -- the checker for the HIP CNN kernels, never the product.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.

*** PARITY UNPINNED ***  The arithmetic is in the external `DeepCNN-MERGED_{mode}.onnx` files (reference
mDeepFRI/__init__.py:68) executed by onnxruntime; neither is available here and no reference test runs them.  Restated is
the published architecture of flatironinstitute/DeepFRI `deepfrier/DeepCNN.py` (no language model; the reference's own
timing table weight_convert/inference_times.csv.gz -- ~14 us per residue on one core -- rules out anything much larger
than ~0.2 M multiply-adds per residue):

    x_b   = Conv1D(filters=F_b, kernel_size=k_b, padding='same')(onehot)      b = 1..n  (parallel branches, bias)
    x     = Concatenate()(x_1..x_n) ; BatchNormalization (inference) ; relu ; Dropout = identity
    g     = GlobalMaxPooling1D()(x)
    z     = g W_out + b_out ; reshape (T,2) ; softmax(-1) ;  y = softmax[:, 0]                       (FuncPredictor)

Keras Conv1D is a cross-correlation; TensorFlow 'same' padding at stride 1 puts (k-1)//2 zeros on the left and the rest
on the right:  x_b[p, f] = bias[f] + sum_j W_b[j, onehot-letter(p + j - (k-1)//2), f].
Weight keys: cnn_W{b} (k_b, 26, F_b), cnn_b{b} (F_b), optional cnn_pad{b} (left padding, default (k_b-1)//2),
bn_gamma / bn_beta / bn_mean / bn_var (sum F_b), bn_eps (scalar, Keras default 1e-3), W_out (sum F_b, 2T), b_out (2T).
"""
import numpy as np

import gcn_oracle


def cnn_forward(weights: dict, seq: str, dtype=np.float32, return_intermediates=False):
    dt = np.dtype(dtype)
    idx = gcn_oracle.seq_to_index(seq)
    L = len(idx)
    feats = []
    b = 1
    while f"cnn_W{b}" in weights:
        W = np.asarray(weights[f"cnn_W{b}"], dtype=dt)
        k, _, F = W.shape
        left = int(np.asarray(weights.get(f"cnn_pad{b}", (k - 1) // 2)).reshape(-1)[0])
        x = np.tile(np.asarray(weights[f"cnn_b{b}"], dtype=dt), (L, 1))
        for j in range(k):
            src = np.arange(L) + j - left
            ok = (src >= 0) & (src < L)
            x[ok] += W[j][idx[src[ok]]]
        feats.append(x)
        b += 1
    x = np.concatenate(feats, axis=1)
    eps = dt.type(np.asarray(weights.get("bn_eps", 1e-3)).reshape(-1)[0])
    g, be = np.asarray(weights["bn_gamma"], dt), np.asarray(weights["bn_beta"], dt)
    mu, var = np.asarray(weights["bn_mean"], dt), np.asarray(weights["bn_var"], dt)
    x = (x - mu) / np.sqrt(var + eps) * g + be
    x = np.maximum(x, 0)
    pooled = x.max(axis=0)
    z = (pooled @ np.asarray(weights["W_out"], dt) + np.asarray(weights["b_out"], dt)).reshape(-1, 2)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    y = (e / e.sum(axis=1, keepdims=True))[:, 0].astype(dt)
    if return_intermediates:
        return y, {"pooled": pooled}
    return y
