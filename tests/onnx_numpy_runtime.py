"""A small ONNX graph interpreter in NumPy (test infrastructure): executes a FILE'S OWN graph, op by op, following the operator
semantics of the ONNX specification (opset 13-15, with the attribute forms of older opsets accepted) -- NOT the oracle's
formulation of the network.  It stands in for onnxruntime, which this image lacks: `run(graph, feeds)` is what
`InferenceSession.run(None, feeds)` would compute (reference predict.pyx:98), in float64 so that it can referee between a graph
and oracle/*.py.  Covered: everything mdfri_testkit.onnx_writer emits, plus the operators a tf2onnx (opset 15) conversion of the
Keras DeepFRI models may leave in a released file -- shape plumbing (Shape / Gather / Slice / Concat / Cast / Expand /
ConstantOfShape / Range / Tile), elementwise and comparison ops, Where, Einsum, reductions, LSTM with optional initial state.
Control flow (Loop / If / Scan) is not: a graph that needs it raises NotImplementedError naming the operator
(tests/validation/validate_release.py prints that as its verdict)."""
import numpy as np


def _lstm(X, W, R, B, hidden, h0=None, c0=None):
    """ONNX LSTM, forward direction, default activations (sigmoid, tanh, tanh), no peepholes.
    X (seq, batch, in); W (1, 4H, in), R (1, 4H, H), B (1, 8H) or None; gate order i, o, f, c; optional initial_h / initial_c
    (1, batch, H).  Returns (Y (seq, 1, batch, H), Y_h (1, batch, H), Y_c (1, batch, H))."""
    W, R = W[0], R[0]
    Wb, Rb = (B[0][:4 * hidden], B[0][4 * hidden:]) if B is not None else (0.0, 0.0)
    seq, batch, _ = X.shape
    h = np.zeros((batch, hidden)) if h0 is None else np.array(h0[0], dtype=np.float64)
    c = np.zeros((batch, hidden)) if c0 is None else np.array(c0[0], dtype=np.float64)
    Y = np.zeros((seq, 1, batch, hidden))
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
    for t in range(seq):
        g = X[t] @ W.T + h @ R.T + Wb + Rb
        i, o, f, cc = (g[:, k * hidden:(k + 1) * hidden] for k in range(4))
        c = sig(f) * c + sig(i) * np.tanh(cc)
        h = sig(o) * np.tanh(c)
        Y[t, 0] = h
    return Y, h[None], c[None]


_NP_OF_ONNX = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


def _axes(node_inputs, attrs, pos=1):
    """axes of Squeeze / Unsqueeze / ReduceX: an input since opset 13, the attribute `axes` before; None = not given."""
    if len(node_inputs) > pos and node_inputs[pos] is not None:
        return tuple(int(i) for i in np.asarray(node_inputs[pos]).reshape(-1))
    if attrs.get("axes") is not None:
        a = attrs["axes"]
        return tuple(int(i) for i in (a if isinstance(a, (list, tuple, np.ndarray)) else [a]))
    return None


def _reduce(fn, x, attrs):
    ax = _axes(x, attrs)
    if ax is None and attrs.get("noop_with_empty_axes", 0):
        return x[0]
    return fn(x[0], axis=ax, keepdims=bool(attrs.get("keepdims", 1)))


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
    """graph: mDeepFRI.onnx_reader.Graph; feeds: {input name: array}.  Returns the list of graph outputs.  Floating-point tensors
    are carried in `dtype`; integer tensors (shapes, indices) keep their type."""
    fl = lambda a: a.astype(dtype) if a.dtype.kind == "f" else a  # noqa: E731
    v = {k: fl(np.asarray(a)) for k, a in graph.initializers.items()}
    v[""] = None                                             # an omitted optional input
    for k, a in feeds.items():
        v[k] = fl(np.asarray(a, dtype=dtype if np.asarray(a).dtype.kind == "f" else None))
    for nd in graph.nodes:
        missing = [i for i in nd.inputs if i not in v]
        if missing:
            raise KeyError(f"node {nd.name or nd.op_type}: input(s) {missing} have no producer (graph not topologically sorted?)")
        x = [v[i] for i in nd.inputs]
        # proto3 does not serialise zero-valued scalars: an attribute that is present without a payload is an integer 0
        a, op = {k: (0 if val is None else val) for k, val in nd.attrs.items()}, nd.op_type
        if op == "MatMul":
            y = np.matmul(x[0], x[1])
        elif op in ("Add", "Sub", "Mul", "Div", "Pow"):
            y = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply, "Div": np.divide, "Pow": np.power}[op](x[0], x[1])
            if op == "Div" and x[0].dtype.kind in "iu" and x[1].dtype.kind in "iu":
                y = (np.trunc(np.true_divide(x[0], x[1]))).astype(x[0].dtype)    # ONNX integer division truncates
        elif op in ("Sqrt", "Exp", "Log", "Tanh", "Abs", "Neg", "Floor", "Ceil", "Reciprocal", "Sign", "Erf"):
            if op == "Erf":
                from math import erf
                y = np.vectorize(erf)(x[0]).astype(dtype)
            else:
                y = {"Sqrt": np.sqrt, "Exp": np.exp, "Log": np.log, "Tanh": np.tanh, "Abs": np.abs, "Neg": np.negative, "Floor": np.floor,
                     "Ceil": np.ceil, "Reciprocal": np.reciprocal, "Sign": np.sign}[op](x[0])
        elif op == "Sigmoid":
            y = 1.0 / (1.0 + np.exp(-x[0]))
        elif op == "Relu":
            y = np.maximum(x[0], 0)
        elif op == "LeakyRelu":
            y = np.where(x[0] > 0, x[0], a.get("alpha", 0.01) * x[0])
        elif op == "Elu":
            al = a.get("alpha", 1.0)
            y = np.where(x[0] > 0, x[0], al * (np.exp(np.minimum(x[0], 0)) - 1.0))
        elif op == "Clip":
            lo = x[1] if len(x) > 1 and x[1] is not None else a.get("min")
            hi = x[2] if len(x) > 2 and x[2] is not None else a.get("max")
            y = np.clip(x[0], lo, hi)
        elif op in ("Max", "Min", "Sum", "Mean"):
            y = x[0]
            for t in x[1:]:
                y = {"Max": np.maximum, "Min": np.minimum, "Sum": np.add, "Mean": np.add}[op](y, t)
            if op == "Mean":
                y = y / len(x)
        elif op in ("Equal", "Less", "Greater", "LessOrEqual", "GreaterOrEqual", "And", "Or"):
            y = {"Equal": np.equal, "Less": np.less, "Greater": np.greater, "LessOrEqual": np.less_equal, "GreaterOrEqual": np.greater_equal,
                 "And": np.logical_and, "Or": np.logical_or}[op](x[0], x[1])
        elif op == "Not":
            y = np.logical_not(x[0])
        elif op == "Where":
            y = np.where(x[0], x[1], x[2])
        elif op in ("Identity", "Dropout"):
            y = x[0]
        elif op == "Cast":
            to = _NP_OF_ONNX[int(a["to"])]
            y = x[0].astype(dtype if np.dtype(to).kind == "f" else to)
        elif op == "Constant":
            val = a.get("value")
            if val is None:
                val = np.asarray(a.get("value_float", a.get("value_int", a.get("value_floats", a.get("value_ints")))))
            y = fl(np.asarray(val))
        elif op == "ConstantOfShape":
            val = a.get("value")
            fill = np.asarray(val).reshape(-1)[0] if val is not None else np.float32(0)
            y = fl(np.full(tuple(int(i) for i in x[0]), fill))
        elif op == "Shape":
            y = np.asarray(x[0].shape, dtype=np.int64)[int(a.get("start", 0)):a.get("end")]
        elif op == "Size":
            y = np.asarray(x[0].size, dtype=np.int64)
        elif op == "Gather":
            y = np.take(x[0], np.asarray(x[1], dtype=np.int64), axis=int(a.get("axis", 0)))
        elif op == "Slice":
            if len(x) > 1:
                starts, ends = x[1], x[2]
                axes = x[3] if len(x) > 3 and x[3] is not None else np.arange(len(starts))
                steps = x[4] if len(x) > 4 and x[4] is not None else np.ones(len(starts), dtype=np.int64)
            else:                                           # opset < 10: attributes
                starts, ends = a["starts"], a["ends"]
                axes, steps = a.get("axes", list(range(len(starts)))), [1] * len(starts)
            sl = [slice(None)] * x[0].ndim
            for s0, e0, ax, st in zip(starts, ends, axes, steps):
                sl[int(ax)] = slice(int(np.clip(s0, -2**62, 2**62)), int(np.clip(e0, -2**62, 2**62)), int(st))
            y = x[0][tuple(sl)]
        elif op == "Expand":
            y = x[0] * np.ones(tuple(int(i) for i in x[1]), dtype=x[0].dtype)
        elif op == "Tile":
            y = np.tile(x[0], tuple(int(i) for i in x[1]))
        elif op == "Range":
            y = np.arange(x[0], x[1], x[2])
        elif op == "EyeLike":
            assert x[0].ndim == 2, "EyeLike takes a 2-D tensor"
            y = np.eye(x[0].shape[0], x[0].shape[1], k=int(a.get("k", 0)), dtype=dtype)
        elif op in ("ReduceSum", "ReduceMax", "ReduceMin", "ReduceMean", "ReduceProd"):
            y = _reduce({"ReduceSum": np.sum, "ReduceMax": np.max, "ReduceMin": np.min, "ReduceMean": np.mean, "ReduceProd": np.prod}[op], x, a)
        elif op == "ReduceL2":
            y = np.sqrt(_reduce(np.sum, [x[0] * x[0]] + x[1:], a))
        elif op == "Einsum":
            y = np.einsum(a["equation"].decode() if isinstance(a["equation"], bytes) else a["equation"], *x)
        elif op == "Squeeze":
            ax = _axes(x, a)
            y = np.squeeze(x[0], axis=ax) if ax is not None else np.squeeze(x[0])
        elif op == "Unsqueeze":
            y = x[0]
            ax = _axes(x, a)
            nd_out = x[0].ndim + len(ax)
            for k in sorted(i % nd_out for i in ax):
                y = np.expand_dims(y, k)
        elif op == "Transpose":
            y = np.transpose(x[0], a.get("perm"))
        elif op == "Concat":
            y = np.concatenate(x, axis=a["axis"])
        elif op == "Split":
            ax = int(a.get("axis", 0))
            sizes = x[1] if len(x) > 1 and x[1] is not None else a.get("split")
            parts = np.split(x[0], np.cumsum([int(i) for i in sizes])[:-1], axis=ax) if sizes is not None else np.split(x[0], len(nd.outputs), axis=ax)
            for name, val in zip(nd.outputs, parts):
                v[name] = val
            continue
        elif op == "Reshape":
            shape = [int(i) for i in x[1]]
            shape = [x[0].shape[k] if (d == 0 and not a.get("allowzero", 0)) else d for k, d in enumerate(shape)]
            y = np.reshape(x[0], tuple(shape))
        elif op == "Flatten":
            ax = a.get("axis", 1)
            y = x[0].reshape(int(np.prod(x[0].shape[:ax])), -1)
        elif op in ("Softmax", "LogSoftmax"):
            e = np.exp(x[0] - np.max(x[0], axis=a.get("axis", -1), keepdims=True))
            y = e / np.sum(e, axis=a.get("axis", -1), keepdims=True)
            if op == "LogSoftmax":
                y = np.log(y)
        elif op == "Gemm":
            A = x[0].T if a.get("transA", 0) else x[0]
            B = x[1].T if a.get("transB", 0) else x[1]
            y = a.get("alpha", 1.0) * (A @ B) + (a.get("beta", 1.0) * x[2] if len(x) > 2 and x[2] is not None else 0.0)
        elif op == "LSTM":
            assert a.get("direction", b"forward") == b"forward", "only forward LSTM nodes are supported"
            assert not a.get("layout", 0), "LSTM layout=1 is not supported"
            assert len(x) <= 4 or x[4] is None, "LSTM sequence_lens is not supported"
            opt = lambda k: x[k] if len(x) > k else None  # noqa: E731
            res = _lstm(x[0], x[1], x[2], opt(3), a["hidden_size"], opt(5), opt(6))
            for name, val in zip(nd.outputs, res):
                if name:
                    v[name] = val
            continue
        elif op == "Conv":
            y = _conv(x[0], x[1], x[2] if len(x) > 2 and x[2] is not None else np.zeros(x[1].shape[0]), a)
        elif op == "BatchNormalization":
            shp = [1, -1] + [1] * (x[0].ndim - 2)
            y = (x[0] - x[3].reshape(shp)) / np.sqrt(x[4].reshape(shp) + a.get("epsilon", 1e-5)) * x[1].reshape(shp) + x[2].reshape(shp)
        elif op == "GlobalMaxPool":
            y = np.max(x[0], axis=tuple(range(2, x[0].ndim)), keepdims=True)
        elif op == "GlobalAveragePool":
            y = np.mean(x[0], axis=tuple(range(2, x[0].ndim)), keepdims=True)
        else:
            raise NotImplementedError(f"ONNX operator {op} (node {nd.name!r}) is not implemented by tests/onnx_numpy_runtime.py")
        v[nd.outputs[0]] = y
    return [v[o] for o in graph.outputs]
