"""The numpy oracles of the floating-point stages against an independent restatement on PyTorch's own CPU operators
(torch.nn.LSTM, F.conv1d, F.batch_norm, F.elu, F.softmax), float64.  The oracles are 'parity unpinned' against the real
model files (none available offline); this guards them against their own mistakes -- gate order, padding side, BatchNorm
form, softmax channel -- with somebody else's implementation of each operator.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cnn_oracle
import gcn_oracle
import lm_oracle
from mdfri_testkit import synthetic


def _t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float64))


def _torch_lstm(x, W, U, b):
    """Keras (I,4H) (H,4H) (4H), gate blocks i,f,c,o == torch.nn.LSTM's (4H,I) (4H,H) blocks i,f,g,o."""
    H = U.shape[0]
    m = torch.nn.LSTM(W.shape[0], H, batch_first=True).double()
    with torch.no_grad():
        m.weight_ih_l0.copy_(_t(W).T)
        m.weight_hh_l0.copy_(_t(U).T)
        m.bias_ih_l0.copy_(_t(b))
        m.bias_hh_l0.zero_()
        return m(_t(x)[None])[0][0].numpy()


def test_lstm_oracle_matches_torch_lstm():
    w = synthetic.glorot_lm_weights(seed=3, hidden=64, embed=256)
    rng = np.random.default_rng(0)
    seq = synthetic.random_sequence(rng, 57)
    S = gcn_oracle.onehot(seq, np.float64)
    h1 = _torch_lstm(S, w["lm_W1"], w["lm_U1"], w["lm_b1"])
    h2 = _torch_lstm(h1, w["lm_W2"], w["lm_U2"], w["lm_b2"])
    np.testing.assert_allclose(lm_oracle.lm_forward(w, seq, dtype=np.float64), h2, atol=1e-12)
    np.testing.assert_allclose(lm_oracle.lm_forward(w, seq, dtype=np.float32), h2, atol=2e-5)


def _torch_gcn(w, seq, A, h2=None):
    S = _t(gcn_oracle.onehot(seq, np.float64))
    A = _t(np.asarray(A, dtype=np.float32))
    x = S @ _t(w["W_aa"])
    if h2 is not None:
        x = x + (_t(h2) @ _t(w["W_lm"]) + _t(w["b_lm"]))
    x = F.relu(x)
    A = A - torch.diag(torch.diag(A)) + torch.eye(A.shape[0], dtype=torch.float64)
    d = 1.0 / (1e-6 + torch.sqrt(A.sum(dim=1)))
    A_hat = torch.diag(d) @ A @ torch.diag(d)
    feats, k = [], 1
    while f"W_gc{k}" in w:
        x = F.elu((A_hat @ x) @ _t(w[f"W_gc{k}"]))
        feats.append(x)
        k += 1
    g = torch.cat(feats, dim=1).sum(dim=0)
    f = F.relu(g @ _t(w["W_fc"]) + _t(w["b_fc"]))
    z = (f @ _t(w["W_out"]) + _t(w["b_out"])).reshape(-1, 2)
    return F.softmax(z, dim=-1)[:, 0].numpy()


@pytest.mark.parametrize("sym", [True, False])
def test_gcn_oracle_matches_torch_ops(sym):
    w = synthetic.glorot_gcn_weights(seed=2, n_terms=23, embed=64, gc_dims=(96, 64, 32), fc_dim=48)
    rng = np.random.default_rng(1)
    seq = synthetic.random_sequence(rng, 41)
    A = rng.integers(0, 2, size=(41, 41)).astype(np.float32)      # the reference notebook's dense random maps
    if sym:
        A = np.maximum(A, A.T)
    np.testing.assert_allclose(gcn_oracle.gcn_forward(w, seq, A, dtype=np.float64), _torch_gcn(w, seq, A), atol=1e-12)
    assert np.abs(gcn_oracle.gcn_forward(w, seq, A) - _torch_gcn(w, seq, A)).max() < 1e-5


def test_gcn_lm_oracle_matches_torch_ops():
    w = synthetic.glorot_gcn_weights(seed=2, n_terms=23, embed=64, gc_dims=(96, 64), fc_dim=48)
    w.update(synthetic.glorot_lm_weights(seed=4, hidden=64, embed=64))
    rng = np.random.default_rng(2)
    seq = synthetic.random_sequence(rng, 33)
    A = (rng.random((33, 33)) < 0.2).astype(np.int32)
    S = gcn_oracle.onehot(seq, np.float64)
    h2 = _torch_lstm(_torch_lstm(S, w["lm_W1"], w["lm_U1"], w["lm_b1"]), w["lm_W2"], w["lm_U2"], w["lm_b2"])
    np.testing.assert_allclose(lm_oracle.gcn_lm_forward(w, seq, A, dtype=np.float64), _torch_gcn(w, seq, A, h2), atol=1e-12)


@pytest.mark.parametrize("L", [1, 3, 19, 20, 64])
def test_cnn_oracle_matches_torch_conv1d(L):
    w = synthetic.glorot_cnn_weights(seed=6, n_terms=15)
    rng = np.random.default_rng(L)
    seq = synthetic.random_sequence(rng, L)
    x = _t(gcn_oracle.onehot(seq, np.float64)).T[None]                 # (1, 26, L)
    outs, b = [], 1
    while f"cnn_W{b}" in w:
        K = _t(w[f"cnn_W{b}"]).permute(2, 1, 0).contiguous()              # Keras (k, C, F) -> torch (F, C, k)
        k = K.shape[2]
        left = (k - 1) // 2                                               # TensorFlow 'same': the extra zero goes to the right
        outs.append(F.conv1d(F.pad(x, (left, k - 1 - left)), K, _t(w[f"cnn_b{b}"])))
        b += 1
    y = torch.cat(outs, dim=1)
    y = F.batch_norm(y, _t(w["bn_mean"]), _t(w["bn_var"]), _t(w["bn_gamma"]), _t(w["bn_beta"]), training=False, eps=float(w["bn_eps"][0]))
    g = F.relu(y).amax(dim=2)[0]
    z = (g @ _t(w["W_out"]) + _t(w["b_out"])).reshape(-1, 2)
    ref = F.softmax(z, dim=-1)[:, 0].numpy()
    np.testing.assert_allclose(cnn_oracle.cnn_forward(w, seq, dtype=np.float64), ref, atol=1e-12)
    assert np.abs(cnn_oracle.cnn_forward(w, seq) - ref).max() < 1e-5
