"""Read the weights of a DeepFRI GCN out of an `.onnx` file without the `onnx` package or onnxruntime.

The reference hands `Predictor` the path of an ONNX file (reference pipeline.py:549-584) that tf2onnx (opset 15) wrote from
the Keras model (reference weight_convert/convert_models2onnx.py:39-43).  This module decodes the protobuf wire format
directly (ModelProto -> GraphProto -> NodeProto / TensorProto; field numbers from the public onnx.proto3) and recovers the
tensors by the STRUCTURE of the graph, not by tensor names (tf2onnx names depend on its version):

    LSTM nodes, in graph order            -> lm_W1/U1/b1, lm_W2/U2/b2   (ONNX gate order i,o,f,c -> Keras order i,f,c,o)
    MatMul/Gemm with a constant (26, E)   -> W_aa       (AA_embedding, no bias)
    MatMul with a constant (H, E) + Add   -> W_lm, b_lm (LM_embedding; only when LSTM nodes exist)
    MatMul chain (E,C1) (C1,C2) (C2,C3)   -> W_gc1..3   (GraphConv kernels, no bias)
    MatMul/Gemm (C1+C2+C3, F) + bias      -> W_fc, b_fc
    MatMul/Gemm (F, 2T) + bias            -> W_out, b_out
  sequence-only DeepCNN files (any Conv node present):
    Conv nodes with constant kernels      -> cnn_W{b} (k,26,F), cnn_b{b}, cnn_pad{b} (from pads / auto_pad)
    BatchNormalization                    -> bn_gamma, bn_beta, bn_mean, bn_var, bn_eps
    MatMul/Gemm (sum F, 2T) + bias        -> W_out, b_out

STATUS: validated against files written with Google's protobuf encoder from the same schema (tests/test_onnx_reader_cpu.py),
NOT against a released DeepFRI file -- none is available offline.  `extract_gcn_weights`
therefore refuses anything it does not recognise instead of guessing.

    python -m mDeepFRI.onnx_reader model.onnx [model.mdfw]      # convert once; Predictor then loads the .mdfw
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field

import numpy as np

__all__ = ["parse_model", "extract_gcn_weights", "extract_cnn_weights", "extract_weights", "load_onnx_weights", "OnnxFormatError"]


class OnnxFormatError(ValueError):
    pass


# ---------------------------------------------------------------------------------------------------------------------
# protobuf wire format
# ---------------------------------------------------------------------------------------------------------------------
def _varint(buf: memoryview, pos: int):
    x, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise OnnxFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        x |= (b & 0x7F) << shift
        if not b & 0x80:
            return x, pos
        shift += 7
        if shift > 70:
            raise OnnxFormatError("varint too long")


def _fields(buf: memoryview):
    """Yield (field_number, wire_type, value) for one message; value is an int (varint / fixed) or a memoryview."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            if pos + ln > n:
                raise OnnxFormatError("truncated length-delimited field")
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            v, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        else:
            raise OnnxFormatError(f"unsupported wire type {wt}")
        yield fno, wt, v


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v, wt):
    if wt == 0:
        return [_signed(v)]
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(_signed(x))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# ONNX messages (only what the extraction needs)
# ---------------------------------------------------------------------------------------------------------------------
_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16,
           11: np.float64, 12: np.uint32, 13: np.uint64}


def _tensor(buf: memoryview):
    """TensorProto: dims=1 data_type=2 float_data=4 int32_data=5 int64_data=7 name=8 raw_data=9 double_data=10
    data_location=14."""
    dims, dtype, name, raw = [], 0, "", None
    f32, i32, i64, f64 = [], [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_varints(v, wt)
        elif fno == 2:
            dtype = v
        elif fno == 4:
            f32.append(np.frombuffer(v, dtype="<f4") if wt == 2 else np.array([struct.unpack("<f", struct.pack("<I", v))[0]], "<f4"))
        elif fno == 5:
            i32 += _packed_varints(v, wt)
        elif fno == 7:
            i64 += _packed_varints(v, wt)
        elif fno == 8:
            name = bytes(v).decode("utf-8", "replace")
        elif fno == 9:
            raw = bytes(v)
        elif fno == 10:
            f64.append(np.frombuffer(v, dtype="<f8") if wt == 2 else np.array([struct.unpack("<d", struct.pack("<Q", v))[0]], "<f8"))
        elif fno == 14 and v == 1:
            raise OnnxFormatError(f"tensor {name!r} uses external data, which is not supported")
    if dtype not in _DTYPES:
        raise OnnxFormatError(f"tensor {name!r}: unsupported data_type {dtype}")
    dt = np.dtype(_DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dtype=dt.newbyteorder("<")).astype(dt)
    elif f32:
        arr = np.concatenate(f32).astype(dt)
    elif f64:
        arr = np.concatenate(f64).astype(dt)
    elif i64:
        arr = np.asarray(i64, dtype=dt)
    elif i32:
        arr = np.asarray(i32).astype(np.uint16).view(np.float16) if dtype == 10 else np.asarray(i32, dtype=dt)
    else:
        arr = np.zeros(0, dtype=dt)
    count = int(np.prod(dims)) if dims else arr.size
    if arr.size != count:
        raise OnnxFormatError(f"tensor {name!r}: {arr.size} values for dims {dims}")
    return name, arr.reshape(dims)


@dataclass
class Node:
    op_type: str = ""
    name: str = ""
    inputs: list = field(default_factory=list)
    outputs: list = field(default_factory=list)
    attrs: dict = field(default_factory=dict)


def _attribute(buf: memoryview):
    """AttributeProto: name=1 f=2 i=3 s=4 t=5 floats=7 ints=8 strings=9."""
    name, val = "", None
    ints, floats, strings = [], [], []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", struct.pack("<I", v))[0]
        elif fno == 3:
            val = _signed(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            floats += list(np.frombuffer(v, dtype="<f4")) if wt == 2 else [struct.unpack("<f", struct.pack("<I", v))[0]]
        elif fno == 8:
            ints += _packed_varints(v, wt)
        elif fno == 9:
            strings.append(bytes(v))
    if val is None:
        val = ints or floats or strings or None
    return name, val


def _node(buf: memoryview) -> Node:
    """NodeProto: input=1 output=2 name=3 op_type=4 attribute=5."""
    nd = Node()
    for fno, wt, v in _fields(buf):
        if fno == 1:
            nd.inputs.append(bytes(v).decode())
        elif fno == 2:
            nd.outputs.append(bytes(v).decode())
        elif fno == 3:
            nd.name = bytes(v).decode()
        elif fno == 4:
            nd.op_type = bytes(v).decode()
        elif fno == 5:
            k, a = _attribute(v)
            nd.attrs[k] = a
    return nd


@dataclass
class Graph:
    nodes: list = field(default_factory=list)
    initializers: dict = field(default_factory=dict)
    inputs: list = field(default_factory=list)
    outputs: list = field(default_factory=list)


def _value_info_name(buf: memoryview) -> str:
    for fno, wt, v in _fields(buf):
        if fno == 1:
            return bytes(v).decode()
    return ""


def _graph(buf: memoryview) -> Graph:
    """GraphProto: node=1 initializer=5 input=11 output=12."""
    g = Graph()
    for fno, wt, v in _fields(buf):
        if fno == 1:
            g.nodes.append(_node(v))
        elif fno == 5:
            name, arr = _tensor(v)
            g.initializers[name] = arr
        elif fno == 11:
            g.inputs.append(_value_info_name(v))
        elif fno == 12:
            g.outputs.append(_value_info_name(v))
    g.inputs = [n for n in g.inputs if n not in g.initializers]   # old exporters list initializers as inputs too
    return g


def parse_model(path_or_bytes) -> Graph:
    """ModelProto: graph=7."""
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    graph = None
    for fno, wt, v in _fields(memoryview(data)):
        if fno == 7 and wt == 2:
            graph = _graph(v)
    if graph is None or not graph.nodes:
        raise OnnxFormatError("no graph found: not an ONNX ModelProto")
    return graph


# ---------------------------------------------------------------------------------------------------------------------
# structural extraction of the DeepFRI GCN tensors
# ---------------------------------------------------------------------------------------------------------------------
_PASS_THROUGH = ("Identity", "Cast", "Squeeze", "Unsqueeze", "Transpose", "Reshape", "Flatten", "Dropout")


def _constants(g: Graph) -> dict:
    """initializers + Constant nodes + constants reached through Identity/Cast (value-preserving for f32 weights)."""
    c = dict(g.initializers)
    for nd in g.nodes:
        if nd.op_type == "Constant" and isinstance(nd.attrs.get("value"), np.ndarray):
            c[nd.outputs[0]] = nd.attrs["value"]
    changed = True
    while changed:
        changed = False
        for nd in g.nodes:
            if nd.op_type in ("Identity", "Cast") and nd.inputs and nd.inputs[0] in c and nd.outputs[0] not in c:
                c[nd.outputs[0]] = c[nd.inputs[0]]
                changed = True
    return c


def _lstm_to_keras(W, R, B, name):
    """ONNX LSTM tensors (1,4H,I) (1,4H,H) (1,8H), gate blocks i,o,f,c -> Keras kernel (I,4H), recurrent (H,4H), bias (4H),
    gate blocks i,f,c,o; bias = Wb + Rb (Keras has one bias vector)."""
    if W.ndim != 3 or W.shape[0] != 1 or R.ndim != 3 or R.shape[0] != 1:
        raise OnnxFormatError(f"{name}: only unidirectional LSTM nodes are supported (W {W.shape}, R {R.shape})")
    H = R.shape[2]
    if W.shape[1] != 4 * H or R.shape[1] != 4 * H:
        raise OnnxFormatError(f"{name}: inconsistent LSTM shapes W {W.shape} R {R.shape}")
    order = (0, 2, 3, 1)  # Keras block k (i,f,c,o) <- ONNX block (i,o,f,c)[order[k]]
    blocks = lambda a: [a[j * H:(j + 1) * H] for j in order]  # noqa: E731
    Wk = np.concatenate(blocks(W[0]), axis=0).T
    Rk = np.concatenate(blocks(R[0]), axis=0).T
    if B is None:
        b = np.zeros(4 * H, np.float32)
    else:
        if B.shape != (1, 8 * H):
            raise OnnxFormatError(f"{name}: LSTM bias has shape {B.shape}, expected (1,{8 * H})")
        b = np.concatenate(blocks(B[0, :4 * H] + B[0, 4 * H:]), axis=0)
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)  # noqa: E731
    return f32(Wk), f32(Rk), f32(b)


def _dense_layers(g: Graph, const: dict):
    """[(weight (in,out), bias or None, node)] for every MatMul / Gemm with one constant 2-D operand, in graph order."""
    consumers = {}
    for nd in g.nodes:
        for i in nd.inputs:
            consumers.setdefault(i, []).append(nd)
    out = []
    for nd in g.nodes:
        if nd.op_type == "MatMul" and len(nd.inputs) == 2 and nd.inputs[1] in const and const[nd.inputs[1]].ndim == 2 \
                and nd.inputs[0] not in const:
            W, bias = const[nd.inputs[1]], None
            y = nd.outputs[0]
            for c in consumers.get(y, []):   # bias: Add(y, const 1-D) directly behind the MatMul
                if c.op_type == "Add":
                    other = [i for i in c.inputs if i != y]
                    if len(other) == 1 and other[0] in const and const[other[0]].size == W.shape[1]:
                        bias = const[other[0]].reshape(-1)
            out.append((np.asarray(W, np.float32), None if bias is None else np.asarray(bias, np.float32), nd))
        elif nd.op_type == "Gemm" and len(nd.inputs) >= 2 and nd.inputs[1] in const and nd.inputs[0] not in const:
            W = const[nd.inputs[1]]
            if nd.attrs.get("transA", 0):
                raise OnnxFormatError(f"Gemm node {nd.name!r} with transA is not supported")
            if nd.attrs.get("transB", 0):
                W = W.T
            W = np.asarray(W, np.float32) * np.float32(nd.attrs.get("alpha", 1.0))
            bias = None
            if len(nd.inputs) == 3 and nd.inputs[2] in const:
                bias = np.asarray(const[nd.inputs[2]], np.float32).reshape(-1) * np.float32(nd.attrs.get("beta", 1.0))
            out.append((W, bias, nd))
    return out


def extract_gcn_weights(g: Graph) -> dict:
    """Graph -> weight dict with the keys of mDeepFRI.weights (validated there).  Raises OnnxFormatError with what was
    found when the graph does not look like a DeepFRI GCN."""
    const = _constants(g)
    w = {}
    lstm = [nd for nd in g.nodes if nd.op_type == "LSTM"]
    if any(nd.op_type == "Loop" for nd in g.nodes) and not lstm:
        raise OnnxFormatError("the recurrent layers were exported as Loop nodes, not LSTM nodes; re-export with a tf2onnx "
                              "version that fuses Keras LSTM layers")
    if len(lstm) not in (0, 2):
        raise OnnxFormatError(f"expected 0 or 2 LSTM nodes (DeepFRI language model), found {len(lstm)}")
    for k, nd in enumerate(lstm, start=1):
        if nd.attrs.get("direction", b"forward") not in (b"forward", "forward"):
            raise OnnxFormatError(f"LSTM node {nd.name!r}: direction {nd.attrs['direction']!r} not supported")
        acts = nd.attrs.get("activations")
        if acts and [a.lower() for a in acts] != [b"sigmoid", b"tanh", b"tanh"]:
            raise OnnxFormatError(f"LSTM node {nd.name!r}: activations {acts} not supported (Sigmoid, Tanh, Tanh expected)")
        ins = nd.inputs + [""] * 8
        if ins[1] not in const or ins[2] not in const:
            raise OnnxFormatError(f"LSTM node {nd.name!r}: W / R are not constants")
        if any(ins[j] for j in (5, 6)) and any(ins[j] in const and np.any(const[ins[j]]) for j in (5, 6)):
            raise OnnxFormatError(f"LSTM node {nd.name!r}: non-zero initial state is not supported")
        B = const.get(ins[3]) if ins[3] else None
        w[f"lm_W{k}"], w[f"lm_U{k}"], w[f"lm_b{k}"] = _lstm_to_keras(const[ins[1]], const[ins[2]], B, nd.name)
    dense = _dense_layers(g, const)
    shapes = [tuple(W.shape) for W, _, _ in dense]

    def fail(msg):
        raise OnnxFormatError(f"{msg}; constant-weight MatMul/Gemm layers found, in graph order: {shapes}")

    used = set()

    def take(pred, what):
        for i, (W, b, nd) in enumerate(dense):
            if i not in used and pred(W, b):
                used.add(i)
                return W, b
        fail(f"no layer matches {what}")

    # AA_embedding: the first constant (26, E) product, with or without a bias
    aa_i = next((i for i, (W, b_, _) in enumerate(dense) if W.shape[0] == 26), None)
    if aa_i is None:
        fail("no layer matches AA_embedding (26, E)")
    producers = {o: nd for nd in g.nodes for o in nd.outputs}
    src = dense[aa_i][2].inputs[0]
    while src in producers and producers[src].op_type in _PASS_THROUGH:
        src = producers[src].inputs[0]
    no_embedding = src in producers and producers[src].op_type == "MatMul" and all(i not in const for i in producers[src].inputs)
    if no_embedding:
        # the (26, C) product sits BEHIND a batch product with the adjacency: it is GraphConv 1 itself and the graph has no
        # embedding layer (the one-hot rows go straight into the stack).  Expressed in the topology the kernels implement as an
        # identity embedding without activation: onehot . I = onehot, exactly.
        w["W_aa"], aa_node = np.eye(26, dtype=np.float32), None
        w["embed_linear"] = np.ones(1, np.float32)
    else:
        used.add(aa_i)
        w["W_aa"], b_aa, aa_node = dense[aa_i]
        if b_aa is not None:
            w["b_aa"] = b_aa
    E = w["W_aa"].shape[1]
    if lstm:
        H = w["lm_U2"].shape[0]
        if w["lm_W1"].shape[0] != 26 or w["lm_W2"].shape[0] != w["lm_U1"].shape[0]:
            fail(f"LSTM shapes {w['lm_W1'].shape} {w['lm_W2'].shape} do not form a 26 -> H -> H stack")
        w["W_lm"], b_lm = take(lambda W, b: W.shape == (H, E), f"LM_embedding ({H}, {E})")
        w["b_lm"] = b_lm if b_lm is not None else np.zeros(E, np.float32)      # upstream puts the bias on one of the two embeddings
    prev, k = E, 0
    while k < 3:
        cand = [i for i, (W, b, _) in enumerate(dense) if i not in used and b is None and W.shape[0] == prev]
        if not cand:
            break
        i = cand[0]
        used.add(i)
        k += 1
        w[f"W_gc{k}"] = dense[i][0]
        prev = dense[i][0].shape[1]
    if k == 0:
        fail(f"no GraphConv kernel ({E}, C) without bias")
    feat = sum(w[f"W_gc{j}"].shape[1] for j in range(1, k + 1))
    w["W_fc"], w["b_fc"] = take(lambda W, b: W.shape[0] == feat and b is not None, f"Dense ({feat}, F) with bias behind the sum pooling")
    F = w["W_fc"].shape[1]
    w["W_out"], w["b_out"] = take(lambda W, b: W.shape[0] == F and b is not None and W.shape[1] % 2 == 0, f"FuncPredictor Dense ({F}, 2T) with bias")
    left = [shapes[i] for i in range(len(dense)) if i not in used]
    if left:
        fail(f"unrecognised extra layers {left} (more than one fully connected layer is not supported)")
    # activation of the embedding: is there a Relu on the way from the AA_embedding product to the first batch product with the
    # adjacency (a MatMul whose other operand is not a constant)?  No Relu -> the `embed_linear` variant.
    consumers = {}
    for nd in g.nodes:
        for i in nd.inputs:
            consumers.setdefault(i, []).append(nd)
    frontier, seen, relu = (list(aa_node.outputs) if aa_node is not None else []), set(), aa_node is None and False
    while frontier and not relu:
        name = frontier.pop()
        for nd in consumers.get(name, []):
            if id(nd) in seen:
                continue
            seen.add(id(nd))
            if nd.op_type == "Relu":
                relu = True
                break
            if nd.op_type == "MatMul" and all(i not in const for i in nd.inputs):
                continue                                   # reached GraphConv's batch_dot(Ahat, X): stop along this path
            if nd.op_type in ("Add", "Identity", "Cast", "Reshape", "Squeeze", "Unsqueeze", "Transpose", "Dropout"):
                frontier += nd.outputs
    if aa_node is not None and not relu:
        w["embed_linear"] = np.ones(1, np.float32)
    return {k_: np.ascontiguousarray(v, dtype=np.float32) for k_, v in w.items()}


def extract_cnn_weights(g: Graph) -> dict:
    """Graph of a sequence-only DeepCNN -> weight dict (keys of mDeepFRI.weights.validate_cnn)."""
    const = _constants(g)
    consumers = {}
    for nd in g.nodes:
        for i in nd.inputs:
            consumers.setdefault(i, []).append(nd)
    w, b = {}, 0
    for nd in g.nodes:
        if nd.op_type != "Conv":
            continue
        if len(nd.inputs) < 2 or nd.inputs[1] not in const:
            raise OnnxFormatError(f"Conv node {nd.name!r}: kernel is not a constant")
        K = np.asarray(const[nd.inputs[1]], np.float32)          # (F, C, k) or a Conv2D form of it, (F, C, 1, k) / (F, C, k, 1)
        spatial = [d for d in K.shape[2:] if d != 1]
        if K.ndim not in (3, 4) or K.shape[1] != 26 or len(spatial) > 1:
            raise OnnxFormatError(f"Conv node {nd.name!r}: kernel shape {K.shape} is not a 1-D convolution over 26 channels")
        k = spatial[0] if spatial else 1
        axis = 0 if K.ndim == 3 else [i for i, d in enumerate(K.shape[2:]) if d == k][-1] if spatial else 0
        K3 = K.reshape(K.shape[0], 26, k)
        if any(v != 1 for v in (nd.attrs.get("strides") or [1])) or any(v != 1 for v in (nd.attrs.get("dilations") or [1])) \
                or nd.attrs.get("group", 1) != 1:
            raise OnnxFormatError(f"Conv node {nd.name!r}: strides / dilations / groups other than 1 are not supported")
        auto = nd.attrs.get("auto_pad", b"NOTSET")
        auto = auto.decode() if isinstance(auto, bytes) else auto
        pads = nd.attrs.get("pads")
        if auto == "SAME_UPPER":
            left = (k - 1) // 2
        elif auto == "SAME_LOWER":
            left = (k - 1) - (k - 1) // 2
        elif pads:
            n_sp = len(pads) // 2
            left, right = int(pads[axis]), int(pads[n_sp + axis])
            if left + right != k - 1:
                raise OnnxFormatError(f"Conv node {nd.name!r}: pads {pads} do not keep the sequence length (kernel {k})")
        elif k == 1:
            left = 0
        else:
            raise OnnxFormatError(f"Conv node {nd.name!r}: 'valid' padding is not the DeepCNN architecture")
        bias = None
        if len(nd.inputs) >= 3 and nd.inputs[2] in const:
            bias = np.asarray(const[nd.inputs[2]], np.float32).reshape(-1)
        else:
            for c in consumers.get(nd.outputs[0], []):
                if c.op_type == "Add":
                    other = [i for i in c.inputs if i != nd.outputs[0]]
                    if len(other) == 1 and other[0] in const and const[other[0]].size == K.shape[0]:
                        bias = np.asarray(const[other[0]], np.float32).reshape(-1)
        b += 1
        w[f"cnn_W{b}"] = np.ascontiguousarray(K3.transpose(2, 1, 0))     # ONNX (F, C, k) -> Keras (k, C, F)
        w[f"cnn_b{b}"] = bias if bias is not None else np.zeros(K.shape[0], np.float32)
        w[f"cnn_pad{b}"] = np.array([left], dtype=np.float32)
    C = sum(w[f"cnn_b{j}"].shape[0] for j in range(1, b + 1))
    bns = [nd for nd in g.nodes if nd.op_type == "BatchNormalization"]
    if len(bns) != 1 or any(i not in const for i in bns[0].inputs[1:5]):
        raise OnnxFormatError(f"expected one BatchNormalization node with constant parameters, found {len(bns)}")
    gamma, beta, mean, var = (np.asarray(const[i], np.float32).reshape(-1) for i in bns[0].inputs[1:5])
    if gamma.size != C:
        raise OnnxFormatError(f"BatchNormalization over {gamma.size} channels, the Conv branches have {C}")
    w["bn_gamma"], w["bn_beta"], w["bn_mean"], w["bn_var"] = gamma, beta, mean, var
    w["bn_eps"] = np.array([bns[0].attrs.get("epsilon", 1e-5)], dtype=np.float32)      # ONNX default 1e-5
    dense = [(W, bias) for W, bias, _ in _dense_layers(g, const) if W.shape[0] == C]
    if len(dense) != 1 or dense[0][1] is None or dense[0][0].shape[1] % 2:
        raise OnnxFormatError(f"expected one FuncPredictor Dense ({C}, 2T) with bias, found {[d[0].shape for d in dense]}")
    w["W_out"], w["b_out"] = dense[0]
    return {k_: np.ascontiguousarray(v, dtype=np.float32) for k_, v in w.items()}


def extract_weights(g: Graph) -> dict:
    """CNN or GCN weight dict, decided by the presence of Conv nodes."""
    return extract_cnn_weights(g) if any(nd.op_type == "Conv" for nd in g.nodes) else extract_gcn_weights(g)


def load_onnx_weights(path: str) -> dict:
    from . import weights as _weights
    w = extract_weights(parse_model(path))
    _weights.validate(w)
    return w


if __name__ == "__main__":
    import sys
    from . import weights as _weights
    if len(sys.argv) not in (2, 3):
        raise SystemExit(__doc__)
    src = sys.argv[1]
    dst = sys.argv[2] if len(sys.argv) == 3 else src.rsplit(".", 1)[0] + ".mdfw"
    wts = load_onnx_weights(src)
    _weights.save_mdfw(dst, wts)
    print(f"{dst}: " + ", ".join(f"{k}{tuple(v.shape)}" for k, v in wts.items()))
