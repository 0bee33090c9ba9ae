"""mDeepFRI.onnx_reader (hand-written protobuf decoding + structural weight extraction) against files written by Google's
protobuf encoder (mdfri_testkit/onnx_writer.py).  CPU only.  No released DeepFRI .onnx file is available offline: what is proven
here is wire-format decoding and the structural mapping on a graph with tf2onnx's op sequence."""
import numpy as np
import pytest

from mdfri_testkit import onnx_writer


def _weights(lm: bool):
    from mdfri_testkit import synthetic
    w = synthetic.glorot_gcn_weights(seed=3, n_terms=17, embed=256, gc_dims=(256, 512, 256), fc_dim=256)
    if lm:
        w.update(synthetic.glorot_lm_weights(seed=5, hidden=64, embed=256))
    return w


@pytest.mark.parametrize("lm", [False, True])
@pytest.mark.parametrize("raw", [True, False])
@pytest.mark.parametrize("gemm", [False, True])
def test_extracts_every_tensor(lm, raw, gemm):
    from mDeepFRI import onnx_reader, weights
    w = _weights(lm)
    g = onnx_reader.parse_model(onnx_writer.deepfri_gcn_model(w, raw=raw, use_gemm_head=gemm))
    assert g.inputs == ["cmap", "seq"]          # the names predict.pyx:82-90 feeds
    got = onnx_reader.extract_gcn_weights(g)
    assert sorted(got) == sorted(w)
    for k in w:
        np.testing.assert_array_equal(got[k], w[k], err_msg=k)
    assert weights.validate(got)["lm_dim"] == (64 if lm else 0)


def test_onnx_path_resolution_and_conversion(tmp_path):
    from mDeepFRI import onnx_reader, weights
    w = _weights(True)
    p = tmp_path / "DeepFRI-MERGED_GraphConv_gcd_512-512-512_fcd_1024_ca_10.0_mf.onnx"
    p.write_bytes(onnx_writer.deepfri_gcn_model(w))
    assert weights.resolve_model_path(str(p)) == str(p)
    got = weights.load_weights(str(p))
    for k in w:
        np.testing.assert_array_equal(got[k], w[k])
    weights.save_mdfw(str(p)[:-5] + ".mdfw", onnx_reader.load_onnx_weights(str(p)))
    assert weights.resolve_model_path(str(p)).endswith(".mdfw")   # a converted sibling wins
    back = weights.load_weights(str(p))
    for k in w:
        np.testing.assert_array_equal(back[k], w[k])


def test_rejects_what_it_does_not_recognise(tmp_path):
    from mDeepFRI import onnx_reader
    with pytest.raises(onnx_reader.OnnxFormatError):
        onnx_reader.parse_model(b"\x00\x01\x02not a protobuf")
    w = _weights(False)
    del w["W_gc2"], w["W_gc3"]
    w["W_fc"] = w["W_fc"][:256]
    b = onnx_writer.GraphBuilder()
    x = b.node("MatMul", [b.input("seq"), b.const(w["W_aa"])])
    b.output(b.node("Relu", [x]))
    with pytest.raises(onnx_reader.OnnxFormatError, match="GraphConv"):
        onnx_reader.extract_gcn_weights(onnx_reader.parse_model(b.serialize()))
    one = onnx_writer.GraphBuilder()                                  # a single LSTM node is not the DeepFRI language model
    Wo, Ro, Bo = onnx_writer.keras_lstm_to_onnx(np.zeros((26, 256), np.float32), np.zeros((64, 256), np.float32), np.zeros(256, np.float32))
    one.output(one.node("LSTM", [one.input("seq"), one.const(Wo), one.const(Ro), one.const(Bo)], n_out=3, hidden_size=64)[0])
    with pytest.raises(onnx_reader.OnnxFormatError, match="LSTM"):
        onnx_reader.extract_gcn_weights(onnx_reader.parse_model(one.serialize()))


def test_lstm_gate_reorder_matches_oracle_semantics():
    """ONNX LSTM blocks are i,o,f,c; the extraction must hand Keras-ordered i,f,c,o tensors to the kernels: run the ONNX
    operator's own definition (iofc) in numpy and compare with oracle/lm_oracle.lstm_forward on the extracted tensors."""
    import lm_oracle
    from mDeepFRI import onnx_reader
    w = _weights(True)
    Wo, Ro, Bo = onnx_writer.keras_lstm_to_onnx(w["lm_W2"], w["lm_U2"], w["lm_b2"])
    Wk, Uk, bk = onnx_reader._lstm_to_keras(Wo, Ro, Bo, "t")
    rng = np.random.default_rng(0)
    x = rng.standard_normal((9, 64)).astype(np.float32)
    H = 64
    h = np.zeros(H, np.float32)
    c = np.zeros(H, np.float32)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))  # noqa: E731
    ref = []
    for t in range(9):   # ONNX LSTM definition, gates i,o,f,c
        z = Wo[0] @ x[t] + Ro[0] @ h + Bo[0, :4 * H] + Bo[0, 4 * H:]
        i, o, f, g = sig(z[:H]), sig(z[H:2 * H]), sig(z[2 * H:3 * H]), np.tanh(z[3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        ref.append(h)
    np.testing.assert_allclose(lm_oracle.lstm_forward(x, Wk, Uk, bk), np.array(ref), atol=1e-6)


@pytest.mark.parametrize("conv2d_form", [False, True])
@pytest.mark.parametrize("explicit_pads", [False, True])
def test_extracts_the_sequence_only_cnn(conv2d_form, explicit_pads):
    from mDeepFRI import onnx_reader, weights
    from mdfri_testkit import synthetic
    w = synthetic.glorot_cnn_weights(seed=4, n_terms=9)
    if explicit_pads:
        w["cnn_pad3"] = np.array([9], dtype=np.float32)      # a non-default split of the 14 padding zeros of kernel 15
    g = onnx_reader.parse_model(onnx_writer.deepcnn_model(w, conv2d_form=conv2d_form, explicit_pads=explicit_pads))
    assert g.inputs == ["seq"]
    got = onnx_reader.extract_weights(g)
    topo = weights.validate(got)
    assert topo["kind"] == "cnn" and topo["kernel_lens"] == [5, 10, 15, 20] and topo["filters"] == [120, 100, 80, 60]
    for k in w:
        np.testing.assert_array_equal(got[k].reshape(-1), np.asarray(w[k], np.float32).reshape(-1), err_msg=k)
    for b, klen in enumerate((5, 10, 15, 20), start=1):
        want = w.get(f"cnn_pad{b}", [(klen - 1) // 2])
        assert int(got[f"cnn_pad{b}"][0]) == int(np.asarray(want).reshape(-1)[0])


# ---- the exported graphs, executed op by op under ONNX operator semantics, against the oracles --------------------------------
def _one_hot(seq):
    import cmap_oracle
    return cmap_oracle.seq2onehot(seq)


@pytest.mark.parametrize("lm", [False, True])
@pytest.mark.parametrize("gemm", [False, True])
def test_exported_gcn_graph_computes_what_the_oracle_states(lm, gemm):
    """What onnxruntime would return for the exported file (tests/onnx_numpy_runtime.py interprets the graph per the ONNX
    operator spec) == oracle/gcn_oracle.py / lm_oracle.py on the same weights and inputs, through the reference's call
    convention (predict.pyx:82-100: A (1,L,L) f32, S (1,L,26) f32, out[0][:, :, 0].reshape(-1)).  This is the check a real ORT
    run of the same file will repeat on a box that has onnxruntime (bench.py probes for it)."""
    import gcn_oracle
    import lm_oracle
    import onnx_numpy_runtime
    from mDeepFRI import onnx_reader
    from mdfri_testkit import synthetic
    w = _weights(lm)
    g = onnx_reader.parse_model(onnx_writer.deepfri_gcn_model(w, use_gemm_head=gemm))
    rng = np.random.default_rng(31)
    for L, dense in ((37, False), (64, True)):
        seq = synthetic.random_sequence(rng, L)
        if dense:   # the reference notebook's recipe: random 0/1, non-symmetric, arbitrary diagonal
            A = rng.integers(0, 2, size=(L, L)).astype(np.float32)
        else:
            import cmap_oracle
            A = cmap_oracle.calculate_contact_map(synthetic.random_walk_coords(rng, L), 6.0).astype(np.float32)
        out = onnx_numpy_runtime.run(g, {"cmap": A.reshape(1, L, L), "seq": _one_hot(seq).reshape(1, L, 26)})
        y = out[0][:, :, 0].reshape(-1)
        ref = (lm_oracle.gcn_lm_forward if lm else gcn_oracle.gcn_forward)(w, seq, A, dtype=np.float64)
        assert y.shape == ref.shape and np.max(np.abs(y - ref)) < 1e-9, (lm, gemm, L)


@pytest.mark.parametrize("conv2d_form,explicit_pads", [(False, False), (True, True)])
def test_exported_cnn_graph_computes_what_the_oracle_states(conv2d_form, explicit_pads):
    import cnn_oracle
    import onnx_numpy_runtime
    from mDeepFRI import onnx_reader
    from mdfri_testkit import synthetic
    w = synthetic.glorot_cnn_weights(seed=2, n_terms=13, filters=(24, 16, 8), kernel_lens=(5, 10, 15))
    g = onnx_reader.parse_model(onnx_writer.deepcnn_model(w, conv2d_form=conv2d_form, explicit_pads=explicit_pads))
    seq = synthetic.random_sequence(np.random.default_rng(3), 41)
    y = onnx_numpy_runtime.run(g, {"seq": _one_hot(seq).reshape(1, len(seq), 26)})[0][:, :, 0].reshape(-1)
    ref = cnn_oracle.cnn_forward(w, seq, dtype=np.float64) if "dtype" in cnn_oracle.cnn_forward.__code__.co_varnames else cnn_oracle.cnn_forward(w, seq)
    assert np.max(np.abs(y - ref)) < 1e-6


@pytest.mark.parametrize("variant", ["embed_linear", "embed_bias", "linear_and_bias", "lm_linear", "no_embedding"])
def test_embedding_topology_variants_are_read_from_the_graph(variant):
    """What a released file's embedding really looks like (activation or not, bias or not, a layer at all or not) cannot be known
    offline, so it is DATA: the reader decides from the graph, the weight dict carries `embed_linear` / `b_aa`, and the file's own
    graph executed under ONNX semantics agrees with the oracle on the extracted tensors."""
    import onnx_numpy_runtime as rt
    import gcn_oracle
    import lm_oracle
    from mDeepFRI import onnx_reader, weights
    from mdfri_testkit import synthetic
    kw = {"embed_linear": dict(embed_linear=True), "embed_bias": dict(embed_bias=True), "linear_and_bias": dict(embed_linear=True, embed_bias=True),
          "lm_linear": dict(embed_linear=True, embed_bias=True), "no_embedding": {}}[variant]
    w = synthetic.glorot_gcn_weights(seed=8, n_terms=13, embed=256, gc_dims=(256, 256), fc_dim=256, **kw)
    if variant == "lm_linear":
        w.update(synthetic.glorot_lm_weights(seed=5, hidden=64, embed=256))
    if variant == "no_embedding":
        del w["W_aa"]
        w["W_gc1"] = synthetic.glorot_uniform(np.random.default_rng(1), 26, 256)
    g = onnx_reader.parse_model(onnx_writer.deepfri_gcn_model(w))
    got = onnx_reader.extract_gcn_weights(g)
    topo = weights.validate(got)
    assert topo["embed_linear"] == (variant != "embed_bias")
    assert ("b_aa" in got) == ("bias" in variant or variant == "lm_linear")
    if variant == "no_embedding":
        assert np.array_equal(got["W_aa"], np.eye(26, dtype=np.float32)) and got["W_gc1"].shape == (26, 256) and topo["embed"] == 26
    else:
        for k in w:
            np.testing.assert_array_equal(got[k], w[k], err_msg=k)
    rng = np.random.default_rng(2)
    seq = synthetic.random_sequence(rng, 50)
    cm = rng.integers(0, 2, size=(50, 50)).astype(np.int32)
    y_graph = rt.run(g, {"cmap": cm.reshape(1, 50, 50).astype(np.float32), "seq": gcn_oracle.onehot(seq).reshape(1, 50, 26)})[0][:, :, 0].reshape(-1)
    fwd = lm_oracle.gcn_lm_forward if variant == "lm_linear" else gcn_oracle.gcn_forward
    assert np.max(np.abs(fwd(got, seq, cm, dtype=np.float64) - y_graph)) < 1e-9
    if variant == "embed_linear":   # and the default graph, executed, differs from it: the flag is not a no-op
        w2 = dict(w)
        del w2["embed_linear"]
        assert np.max(np.abs(gcn_oracle.gcn_forward(w2, seq, cm, dtype=np.float64) - y_graph)) > 1e-6
