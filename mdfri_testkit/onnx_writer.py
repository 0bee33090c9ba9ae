"""Export weight dicts (mDeepFRI.weights) as ONNX ModelProto files: the op sequence of a tf2onnx export of the DeepFRI GCN /
DeepCNN Keras models, with typed graph inputs/outputs, so that the files are (a) what mDeepFRI.onnx_reader must be able to
read back and (b) runnable by a real onnxruntime -- bench.py does exactly that whenever `import onnxruntime` succeeds on the
box, which makes ORT-CPU on these files the reference-configuration leg of the measurement (reference predict.pyx:62-73,98)
and the first external check of oracle/gcn_oracle.py.  No `onnx` package is needed: the messages are encoded with Google's
protobuf runtime from a schema declared here (the public onnx.proto3 field numbers).  Independent of mDeepFRI/onnx_reader.py,
which decodes the wire format by hand."""
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory

_F = descriptor_pb2.FieldDescriptorProto


def _schema():
    fd = descriptor_pb2.FileDescriptorProto()
    fd.name = "onnx_subset.proto"
    fd.package = "onnxsub"
    fd.syntax = "proto3"

    def msg(name, fields):
        m = fd.message_type.add()
        m.name = name
        for fname, num, ftype, label, tname in fields:
            f = m.field.add()
            f.name, f.number, f.type, f.label = fname, num, ftype, label
            if tname:
                f.type_name = ".onnxsub." + tname
    O, R = _F.LABEL_OPTIONAL, _F.LABEL_REPEATED
    msg("TensorProto", [("dims", 1, _F.TYPE_INT64, R, None), ("data_type", 2, _F.TYPE_INT32, O, None),
                        ("float_data", 4, _F.TYPE_FLOAT, R, None), ("int64_data", 7, _F.TYPE_INT64, R, None),
                        ("name", 8, _F.TYPE_STRING, O, None), ("raw_data", 9, _F.TYPE_BYTES, O, None)])
    msg("AttributeProto", [("name", 1, _F.TYPE_STRING, O, None), ("f", 2, _F.TYPE_FLOAT, O, None), ("i", 3, _F.TYPE_INT64, O, None),
                           ("s", 4, _F.TYPE_BYTES, O, None), ("t", 5, _F.TYPE_MESSAGE, O, "TensorProto"),
                           ("floats", 7, _F.TYPE_FLOAT, R, None), ("ints", 8, _F.TYPE_INT64, R, None),
                           ("strings", 9, _F.TYPE_BYTES, R, None), ("type", 20, _F.TYPE_INT32, O, None)])
    msg("NodeProto", [("input", 1, _F.TYPE_STRING, R, None), ("output", 2, _F.TYPE_STRING, R, None), ("name", 3, _F.TYPE_STRING, O, None),
                      ("op_type", 4, _F.TYPE_STRING, O, None), ("attribute", 5, _F.TYPE_MESSAGE, R, "AttributeProto")])
    msg("Dimension", [("dim_value", 1, _F.TYPE_INT64, O, None), ("dim_param", 2, _F.TYPE_STRING, O, None)])
    msg("TensorShapeProto", [("dim", 1, _F.TYPE_MESSAGE, R, "Dimension")])
    msg("TypeTensor", [("elem_type", 1, _F.TYPE_INT32, O, None), ("shape", 2, _F.TYPE_MESSAGE, O, "TensorShapeProto")])
    msg("TypeProto", [("tensor_type", 1, _F.TYPE_MESSAGE, O, "TypeTensor")])
    msg("ValueInfoProto", [("name", 1, _F.TYPE_STRING, O, None), ("type", 2, _F.TYPE_MESSAGE, O, "TypeProto")])
    msg("GraphProto", [("node", 1, _F.TYPE_MESSAGE, R, "NodeProto"), ("name", 2, _F.TYPE_STRING, O, None),
                       ("initializer", 5, _F.TYPE_MESSAGE, R, "TensorProto"), ("input", 11, _F.TYPE_MESSAGE, R, "ValueInfoProto"),
                       ("output", 12, _F.TYPE_MESSAGE, R, "ValueInfoProto")])
    msg("OperatorSetIdProto", [("domain", 1, _F.TYPE_STRING, O, None), ("version", 2, _F.TYPE_INT64, O, None)])
    msg("ModelProto", [("ir_version", 1, _F.TYPE_INT64, O, None), ("producer_name", 2, _F.TYPE_STRING, O, None),
                       ("graph", 7, _F.TYPE_MESSAGE, O, "GraphProto"), ("opset_import", 8, _F.TYPE_MESSAGE, R, "OperatorSetIdProto")])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return {n: message_factory.GetMessageClass(pool.FindMessageTypeByName("onnxsub." + n))
            for n in ("TensorProto", "AttributeProto", "NodeProto", "ValueInfoProto", "GraphProto", "ModelProto")}


_M = _schema()


class GraphBuilder:
    def __init__(self):
        self.model = _M["ModelProto"]()
        self.model.ir_version = 8
        self.model.producer_name = "tf2onnx"
        op = self.model.opset_import.add()
        op.version = 15
        self.g = self.model.graph
        self.g.name = "tf2onnx"
        self._n = 0

    def fresh(self, stem):
        self._n += 1
        return f"{stem}:{self._n}"

    @staticmethod
    def _typed(vi, name, shape):
        """float32 tensor value info; `shape` entries are ints or symbolic names (TypeProto.Tensor, onnx.proto3)."""
        vi.name = name
        if shape is not None:
            vi.type.tensor_type.elem_type = 1
            for d in shape:
                dim = vi.type.tensor_type.shape.dim.add()
                if isinstance(d, str):
                    dim.dim_param = d
                else:
                    dim.dim_value = int(d)

    def input(self, name, shape=None):
        self._typed(self.g.input.add(), name, shape)
        return name

    def output(self, name, shape=None):
        self._typed(self.g.output.add(), name, shape)

    def const(self, arr, stem="const", raw=True):
        arr = np.asarray(arr)
        t = self.g.initializer.add()
        t.name = self.fresh(stem)
        t.dims.extend(arr.shape)
        if arr.dtype == np.float32:
            t.data_type = 1
            if raw:
                t.raw_data = arr.astype("<f4").tobytes()
            else:
                t.float_data.extend(arr.reshape(-1).tolist())
        elif arr.dtype == np.int64:
            t.data_type = 7
            if raw:
                t.raw_data = arr.astype("<i8").tobytes()
            else:
                t.int64_data.extend(arr.reshape(-1).tolist())
        else:
            raise TypeError(arr.dtype)
        return t.name

    def node(self, op, inputs, n_out=1, **attrs):
        nd = self.g.node.add()
        nd.op_type = op
        nd.name = self.fresh(op)
        nd.input.extend(inputs)
        outs = [self.fresh(op + "_out") for _ in range(n_out)]
        nd.output.extend(outs)
        for k, v in attrs.items():
            a = nd.attribute.add()
            a.name = k
            if isinstance(v, float):
                a.f, a.type = v, 1
            elif isinstance(v, int):
                a.i, a.type = v, 2
            elif isinstance(v, bytes):
                a.s, a.type = v, 3
            elif isinstance(v, (list, tuple)) and v and isinstance(v[0], bytes):
                a.strings.extend(v)
                a.type = 8
            elif isinstance(v, (list, tuple)):
                a.ints.extend(v)
                a.type = 7
            else:
                raise TypeError(k)
        return outs[0] if n_out == 1 else outs

    def serialize(self) -> bytes:
        return self.model.SerializeToString()


def keras_lstm_to_onnx(W, U, b):
    """Keras (I,4H) (H,4H) (4H) in i,f,c,o order -> ONNX W (1,4H,I), R (1,4H,H), B (1,8H) in i,o,f,c order (recurrent bias zero)."""
    H = U.shape[0]
    blk = lambda a: [a[..., j * H:(j + 1) * H] for j in range(4)]  # noqa: E731
    i, f, c, o = blk(W)
    Wo = np.concatenate([i, o, f, c], axis=1).T[None]
    i, f, c, o = blk(U)
    Ro = np.concatenate([i, o, f, c], axis=1).T[None]
    i, f, c, o = blk(b)
    Bo = np.concatenate([i, o, f, c, np.zeros(4 * H, np.float32)])[None]
    return Wo.astype(np.float32), Ro.astype(np.float32), Bo.astype(np.float32)


def deepfri_gcn_model(w: dict, raw=True, use_gemm_head=False) -> bytes:
    """An ONNX graph with the op sequence tf2onnx emits for the DeepFRI GCN (Keras functional model: LSTM language model,
    AA/LM embeddings, GraphConv layers with adjacency normalisation, sum pooling, dense head, pair softmax)."""
    b = GraphBuilder()
    cmap, seq = b.input("cmap", [1, "L", "L"]), b.input("seq", [1, "L", 26])   # the two feeds of predict.pyx:82-90
    if "W_aa" not in w:      # topology variant: no embedding layer at all, GraphConv 1 takes the one-hot rows (W_gc1 is (26, C))
        x_aa = seq
    else:
        x_aa = b.node("MatMul", [seq, b.const(w["W_aa"], "AA_embedding/kernel", raw)])
    if "b_aa" in w:
        x_aa = b.node("Add", [x_aa, b.const(w["b_aa"], "AA_embedding/bias", raw)])
    if "lm_W1" in w:
        h = b.node("Transpose", [seq], perm=[1, 0, 2])
        for k in (1, 2):
            Wo, Ro, Bo = keras_lstm_to_onnx(w[f"lm_W{k}"], w[f"lm_U{k}"], w[f"lm_b{k}"])
            y = b.node("LSTM", [h, b.const(Wo, f"LSTM{k}/W", raw), b.const(Ro, f"LSTM{k}/R", raw), b.const(Bo, f"LSTM{k}/B", raw)],
                       n_out=3, hidden_size=int(Ro.shape[2]), direction=b"forward")[0]
            h = b.node("Squeeze", [y, b.const(np.array([1], np.int64), "axes", raw)])
        h = b.node("Transpose", [h], perm=[1, 0, 2])
        x_lm = b.node("Add", [b.node("MatMul", [h, b.const(w["W_lm"], "LM_embedding/kernel", raw)]), b.const(w["b_lm"], "LM_embedding/bias", raw)])
        x_aa = b.node("Add", [x_lm, x_aa])
    linear = "W_aa" not in w or ("embed_linear" in w and float(np.asarray(w["embed_linear"]).reshape(-1)[0]) != 0.0)
    x = x_aa if linear else b.node("Relu", [x_aa])
    # adjacency normalisation of GraphConv, arithmetically complete (constants that are NOT weights -- eps, one, axes -- must be
    # ignored by the reader):  A' = A - A*I + I;  d = 1 / (sqrt(rowsum A') + 1e-6);  Ahat = diag(d) A' diag(d)
    a2 = b.node("Squeeze", [cmap, b.const(np.array([0], np.int64), "axes", raw)])                   # (L, L): EyeLike wants rank 2
    eye = b.node("EyeLike", [a2])
    a_hat = b.node("Add", [b.node("Sub", [a2, b.node("Mul", [a2, eye])]), eye])
    deg = b.node("ReduceSum", [a_hat, b.const(np.array([1], np.int64), "axes", raw)], keepdims=1)    # (L, 1)
    dinv = b.node("Div", [b.const(np.array(1.0, np.float32), "one", raw),
                          b.node("Add", [b.node("Sqrt", [deg]), b.const(np.array(1e-6, np.float32), "eps", raw)])])
    a_norm = b.node("Mul", [b.node("Mul", [a_hat, dinv]), b.node("Transpose", [dinv], perm=[1, 0])])  # rows, then columns
    a_norm = b.node("Unsqueeze", [a_norm, b.const(np.array([0], np.int64), "axes", raw)])            # (1, L, L)
    feats, k = [], 1
    while f"W_gc{k}" in w:
        ax = b.node("MatMul", [a_norm, x])                                                           # batch_dot(Ahat, H)
        x = b.node("Elu", [b.node("MatMul", [ax, b.const(w[f"W_gc{k}"], f"GraphConv_{k}/kernel", raw)])], alpha=1.0)
        feats.append(x)
        k += 1
    cat = b.node("Concat", feats, axis=2) if len(feats) > 1 else feats[0]
    pooled = b.node("ReduceSum", [cat, b.const(np.array([1], np.int64), "axes", raw)], keepdims=0)
    if use_gemm_head:
        f = b.node("Relu", [b.node("Gemm", [pooled, b.const(w["W_fc"].T.copy(), "dense/kernel", raw), b.const(w["b_fc"], "dense/bias", raw)], transB=1)])
    else:
        f = b.node("Relu", [b.node("Add", [b.node("MatMul", [pooled, b.const(w["W_fc"], "dense/kernel", raw)]), b.const(w["b_fc"], "dense/bias", raw)])])
    z = b.node("Add", [b.node("MatMul", [f, b.const(w["W_out"], "labels/kernel", raw)]), b.const(w["b_out"], "labels/bias", raw)])
    z = b.node("Reshape", [z, b.const(np.array([-1, w["W_out"].shape[1] // 2, 2], np.int64), "shape", raw)])
    b.output(b.node("Softmax", [z], axis=-1), [1, w["W_out"].shape[1] // 2, 2])
    return b.serialize()


def deepcnn_model(w: dict, conv2d_form=False, explicit_pads=False) -> bytes:
    """An ONNX graph with the op sequence tf2onnx emits for the sequence-only DeepCNN (parallel Conv1D branches over the
    one-hot sequence, concat, BatchNormalization, relu, global max pool, FuncPredictor).  conv2d_form: kernels as (F,C,1,k)
    the way tf2onnx lowers Conv1D through Conv2D; explicit_pads: `pads` attribute instead of auto_pad=SAME_UPPER."""
    b = GraphBuilder()
    seq = b.input("seq", [1, "L", 26])
    x = b.node("Transpose", [seq], perm=[0, 2, 1])
    if conv2d_form:
        x = b.node("Unsqueeze", [x, b.const(np.array([2], np.int64), "axes")])
    branches, k = [], 1
    while f"cnn_W{k}" in w:
        W = w[f"cnn_W{k}"].transpose(2, 1, 0)                       # Keras (k, C, F) -> ONNX (F, C, k)
        klen = W.shape[2]
        left = int(np.asarray(w.get(f"cnn_pad{k}", (klen - 1) // 2)).reshape(-1)[0])
        if conv2d_form:
            W = W[:, :, None, :]
        attrs = {"kernel_shape": ([1, klen] if conv2d_form else [klen])}
        if explicit_pads:
            attrs["pads"] = [0, left, 0, klen - 1 - left] if conv2d_form else [left, klen - 1 - left]
        else:
            attrs["auto_pad"] = b"SAME_UPPER"
        branches.append(b.node("Conv", [x, b.const(np.ascontiguousarray(W), f"conv1d_{k}/kernel"), b.const(w[f"cnn_b{k}"], f"conv1d_{k}/bias")], **attrs))
        k += 1
    cat = b.node("Concat", branches, axis=1)
    bn = b.node("BatchNormalization", [cat, b.const(w["bn_gamma"], "bn/gamma"), b.const(w["bn_beta"], "bn/beta"),
                                       b.const(w["bn_mean"], "bn/mean"), b.const(w["bn_var"], "bn/var")],
                epsilon=float(np.asarray(w["bn_eps"]).reshape(-1)[0]))
    pooled = b.node("GlobalMaxPool", [b.node("Relu", [bn])])
    pooled = b.node("Flatten", [pooled], axis=1)
    z = b.node("Add", [b.node("MatMul", [pooled, b.const(w["W_out"], "labels/kernel")]), b.const(w["b_out"], "labels/bias")])
    z = b.node("Reshape", [z, b.const(np.array([-1, w["W_out"].shape[1] // 2, 2], np.int64), "shape")])
    b.output(b.node("Softmax", [z], axis=-1), [1, w["W_out"].shape[1] // 2, 2])
    return b.serialize()
