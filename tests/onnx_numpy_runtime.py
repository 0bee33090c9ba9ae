"""A minimal ONNX graph interpreter in NumPy (test infrastructure): executes the graphs mDeepFRI.onnx_writer exports, op by op,
following the operator semantics of the ONNX specification (opset 15) -- NOT the oracle's formulation of the network.  It
stands in for onnxruntime, which this image lacks: `run(graph, feeds)` is what `InferenceSession.run(None, feeds)` would
compute (reference predict.pyx:98), in float64 so that it can referee between the exported graph and oracle/*.py.
Only the operators the exported graphs use are implemented; anything else raises."""
import numpy as np


def _lstm(X, W, R, B, hidden):
    """ONNX LSTM, forward direction, default activations (sigmoid, tanh, tanh), no peepholes, zero initial state.
    X (seq, batch, in); W (1, 4H, in), R (1, 4H, H), B (1, 8H); gate order i, o, f, c.  Returns Y (seq, 1, batch, H)."""
    W, R = W[0], R[0]
    Wb, Rb = B[0][:4 * hidden], B[0][4 * hidden:]
    seq, batch, _ = X.shape
    h = np.zeros((batch, hidden))
    c = np.zeros((batch, hidden))
    Y = np.zeros((seq, 1, batch, hidden))
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
    for t in range(seq):
        g = X[t] @ W.T + h @ R.T + Wb + Rb
        i, o, f, cc = (g[:, k * hidden:(k + 1) * hidden] for k in range(4))
        c = sig(f) * c + sig(i) * np.tanh(cc)
        h = sig(o) * np.tanh(c)
        Y[t, 0] = h
    return Y


def _conv(X, W, b, attrs):
    """ONNX Conv for (N, C, L) or (N, C, 1, L) inputs, stride 1, auto_pad SAME_UPPER or explicit pads."""
    two_d = X.ndim == 4
    if two_d:
        X, W = X[:, :, 0, :], W[:, :, 0, :]
    k = W.shape[2]
    if attrs.get("auto_pad") == b"SAME_UPPER":
        total = k - 1
        left, right = total // 2, total - total // 2       # SAME_UPPER: the extra padding goes at the end
    else:
        pads = attrs.get("pads", [0, 0])
        left, right = (pads[1], pads[3]) if two_d else (pads[0], pads[1])
    Xp = np.pad(X, ((0, 0), (0, 0), (left, right)))
    L = Xp.shape[2] - k + 1
    out = np.zeros((X.shape[0], W.shape[0], L))
    for j in range(k):                                      # cross-correlation, as ONNX defines Conv
        out += np.einsum("ncl,fc->nfl", Xp[:, :, j:j + L], W[:, :, j])
    out += b[None, :, None]
    return out[:, :, None, :] if two_d else out


def run(graph, feeds, dtype=np.float64):
    """graph: mDeepFRI.onnx_reader.Graph; feeds: {input name: array}.  Returns the list of graph outputs."""
    v = {k: (a.astype(dtype) if a.dtype.kind == "f" else a) for k, a in graph.initializers.items()}
    for k, a in feeds.items():
        v[k] = np.asarray(a, dtype=dtype)
    for nd in graph.nodes:
        x = [v[i] for i in nd.inputs]
        a, op = nd.attrs, nd.op_type
        if op == "MatMul":
            y = np.matmul(x[0], x[1])
        elif op in ("Add", "Sub", "Mul", "Div"):
            y = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply, "Div": np.divide}[op](x[0], x[1])
        elif op == "Sqrt":
            y = np.sqrt(x[0])
        elif op == "Relu":
            y = np.maximum(x[0], 0)
        elif op == "Elu":
            al = a.get("alpha", 1.0)
            y = np.where(x[0] > 0, x[0], al * (np.exp(np.minimum(x[0], 0)) - 1.0))
        elif op == "EyeLike":
            assert x[0].ndim == 2, "EyeLike takes a 2-D tensor"
            y = np.eye(x[0].shape[0], x[0].shape[1], dtype=dtype)
        elif op == "ReduceSum":
            y = np.sum(x[0], axis=tuple(int(i) for i in x[1]), keepdims=bool(a.get("keepdims", 1)))
        elif op == "Squeeze":
            y = np.squeeze(x[0], axis=tuple(int(i) for i in x[1]))
        elif op == "Unsqueeze":
            y = x[0]
            for ax in sorted(int(i) for i in x[1]):
                y = np.expand_dims(y, ax)
        elif op == "Transpose":
            y = np.transpose(x[0], a["perm"])
        elif op == "Concat":
            y = np.concatenate(x, axis=a["axis"])
        elif op == "Reshape":
            y = np.reshape(x[0], tuple(int(i) for i in x[1]))
        elif op == "Flatten":
            ax = a.get("axis", 1)
            y = x[0].reshape(int(np.prod(x[0].shape[:ax])), -1)
        elif op == "Softmax":
            e = np.exp(x[0] - np.max(x[0], axis=a.get("axis", -1), keepdims=True))
            y = e / np.sum(e, axis=a.get("axis", -1), keepdims=True)
        elif op == "Gemm":
            A = x[0].T if a.get("transA", 0) else x[0]
            B = x[1].T if a.get("transB", 0) else x[1]
            y = a.get("alpha", 1.0) * (A @ B) + (a.get("beta", 1.0) * x[2] if len(x) > 2 else 0.0)
        elif op == "LSTM":
            assert a.get("direction", b"forward") == b"forward"
            Y = _lstm(x[0], x[1], x[2], x[3], a["hidden_size"])
            for name, val in zip(nd.outputs, (Y, Y[-1], None)):
                if name and val is not None:
                    v[name] = val
            continue
        elif op == "Conv":
            y = _conv(x[0], x[1], x[2], a)
        elif op == "BatchNormalization":
            shp = [1, -1] + [1] * (x[0].ndim - 2)
            y = (x[0] - x[3].reshape(shp)) / np.sqrt(x[4].reshape(shp) + a.get("epsilon", 1e-5)) * x[1].reshape(shp) + x[2].reshape(shp)
        elif op == "GlobalMaxPool":
            y = np.max(x[0], axis=tuple(range(2, x[0].ndim)), keepdims=True)
        else:
            raise NotImplementedError(f"ONNX operator {op} is not part of the exported graphs")
        v[nd.outputs[0]] = y
    return [v[o] for o in graph.outputs]
