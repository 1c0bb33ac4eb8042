"""reading the weights out of a shipped model.onnx.  No real vosk model is available
offline, so the wire parser and the name/shape -> hparams mapping are exercised on a synthetic ONNX file
written by an independent minimal protobuf writer (initializers under the reference's state_dict names)."""
import os

import numpy as np
import pytest


def test_roundtrip_synthetic_onnx(tmp_path):
    from vosk_tts_amd import onnx_import as oi
    from vosk_tts_amd import weights as W

    hp = W.tiny_hparams()
    tens = W.make_synthetic_weights(hp, 99)
    extra = dict(tens)
    extra["dec.stft.inverse_basis"] = np.zeros((18, 1, 16), np.float32)  # buffers the exporter also writes
    extra["onnx::Conv_1234"] = np.ones((3, 3), np.float32)             # anonymous constants are ignored
    path = oi.write_minimal_onnx(str(tmp_path / "model.onnx"), extra, use_float_data={"enc_p.proj.bias", "dp.flows.0.m"})
    got_hp, got = oi.import_onnx(path)
    for f, _ in W.HParams._fields_:
        if f in ("reserved", "up_rates", "up_kernels", "res_kernels", "res_dilations"):
            continue
        assert getattr(got_hp, f) == pytest.approx(getattr(hp, f)), f
    assert list(got_hp.up_kernels)[:2] == [16, 16] and list(got_hp.res_kernels)[:3] == [3, 7, 11]
    assert set(got) == set(tens)
    assert all(np.array_equal(got[k], tens[k]) for k in tens)
    # and the produced blob is what the engine/oracle parse
    blob_path = str(tmp_path / "model.vitsw")
    oi.convert(path, blob_path, {"upsample_rates": [4, 4], "resblock_dilation_sizes": [[1, 3, 5]] * 3})
    hp2, t2 = W.unpack_blob(open(blob_path, "rb").read())
    assert hp2.hidden_channels == hp.hidden_channels and np.array_equal(t2["emb_g.weight"], tens["emb_g.weight"])


def test_imported_blob_runs_in_the_oracle(tmp_path, oracle_lib, oracle_tiny):
    from vosk_tts_amd import onnx_import as oi
    from vosk_tts_amd import weights as W

    hp = W.tiny_hparams()
    tens = W.make_synthetic_weights(hp, 1234)
    path = oi.write_minimal_onnx(str(tmp_path / "m.onnx"), tens)
    oi.convert(path, str(tmp_path / "m.vitsw"))
    m = oracle_lib.create(open(tmp_path / "m.vitsw", "rb").read())
    ids = np.array([[1, 5, 0, 7, 3]], np.int64)
    a, _ = m.synthesize(ids, [5], [0.0, 1.0, 0.0], [1], forced_durations=np.full((1, 5), 2, np.int32))
    b, _ = oracle_tiny.synthesize(ids, [5], [0.0, 1.0, 0.0], [1], forced_durations=np.full((1, 5), 2, np.int32))
    assert np.array_equal(a, b)


def test_foreign_graphs_are_rejected_loudly(tmp_path):
    from vosk_tts_amd import onnx_import as oi
    from vosk_tts_amd import weights as W

    tens = W.make_synthetic_weights(W.tiny_hparams(), 1)
    missing = {k: v for k, v in tens.items() if not k.startswith("dp.")}
    with pytest.raises(NotImplementedError, match="dp.pre.weight"):
        oi.import_onnx(oi.write_minimal_onnx(str(tmp_path / "a.onnx"), missing))
    bert = dict(tens, **{"bert_proj.weight": np.zeros((4, 4), np.float32)})
    with pytest.raises(NotImplementedError, match="BERT"):
        oi.import_onnx(oi.write_minimal_onnx(str(tmp_path / "b.onnx"), bert))
    bad = dict(tens)
    bad["dec.conv_pre.bias"] = np.zeros(7, np.float32)
    with pytest.raises(NotImplementedError, match="unexpected shapes"):
        oi.import_onnx(oi.write_minimal_onnx(str(tmp_path / "c.onnx"), bad))
    with pytest.raises(ValueError):
        oi.read_initializers(b"\x3a\xff\xff\xff\xff\x0f")  # length-delimited field running past the buffer


def test_varint_and_negative_int64_fields():
    from vosk_tts_amd import onnx_import as oi

    # TensorProto{dims:[2], data_type: INT64(7), int64_data packed [-1, 300], name "x"}
    neg1 = oi._enc_varint((1 << 64) - 1)
    tp = oi._enc_field(1, 0, 2) + oi._enc_field(2, 0, 7) + oi._enc_field(7, 2, neg1 + oi._enc_varint(300)) + oi._enc_field(8, 2, b"x")
    model = oi._enc_field(7, 2, oi._enc_field(5, 2, tp))
    t = oi.read_initializers(model)
    assert t["x"].tolist() == [-1, 300]


def test_exporter_renamings_are_undone(tmp_path):
    """the three things the real exporter does to parameter names (see OnnxGraph), rebuilt with the minimal writer so the check
    also runs where the reference is absent: Identity de-duplication, Linear weight -> anonymous transposed MatMul constant,
    ElementwiseAffine logs -> anonymous -logs feeding Exp"""
    from vosk_tts_amd import onnx_import as oi
    from vosk_tts_amd import weights as W

    hp = W.tiny_hparams()
    tens = W.make_synthetic_weights(hp, 5)
    g1, g0 = "enc_p.encoder.norm_layers_1.1.gamma", "enc_p.encoder.norm_layers_1.0.gamma"
    tens[g1] = tens[g0].copy()
    f = dict(tens)
    nodes = [("Identity_7", "Identity", [g0], [g1])]
    del f[g1]
    f["onnx::MatMul_901"] = np.ascontiguousarray(f.pop("enc_p.encoder.spk_emb_linear.weight").T)
    nodes.append(("/enc_p/encoder/spk_emb_linear/MatMul", "MatMul", ["/enc_p/x", "onnx::MatMul_901"], ["/enc_p/y"]))
    f["onnx::Exp_902"] = -f.pop("dp.flows.0.logs")
    nodes.append(("/dp/flows.0/Exp", "Exp", ["onnx::Exp_902"], ["/dp/flows.0/Exp_output_0"]))
    hp2, got = oi.import_onnx(oi.write_minimal_onnx(str(tmp_path / "m.onnx"), f, nodes=nodes))
    assert hp2.enc_cond_layer == 2 and set(got) == set(tens)
    assert all(np.array_equal(got[k], tens[k]) for k in tens)
    # an exporter that does not name its nodes: the anonymous constants are matched by shape when unique
    anon = [(f"{op}_{i}", op, ins, outs) for i, (_n, op, ins, outs) in enumerate(nodes)]
    hp3, got3 = oi.import_onnx(oi.write_minimal_onnx(str(tmp_path / "n.onnx"), f, nodes=anon))
    assert all(np.array_equal(got3[k], tens[k]) for k in tens)


def _reference_present():
    return os.path.isdir("/root/reference/training/vits2")


@pytest.mark.skipif(not _reference_present(), reason="container-only: runs the reference's own ONNX export procedure")
def test_real_torch_export_of_the_reference_model_imports_bit_exactly():
    """The REAL exporter output, not a synthetic file: the reference SynthesizerTrn (tiny widths, synthetic weights) goes through
    training/vits2/onnx_export.py's procedure (oracle/onnx_export_ref.py), and the importer must give back the byte-identical
    VITSW001 blob.  What the exporter does to names (observed): spk_emb_linear.weight is folded into a transposed anonymous
    MatMul constant, dp.flows.0.logs into an anonymous -logs feeding Exp, and byte-identical tensors are de-duplicated behind
    Identity nodes - the second model below has untrained-style repeated tensors to exercise exactly that."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import gen_golden as G
    import onnx_export_ref as X

    from vosk_tts_amd import onnx_import as oi
    from vosk_tts_amd import weights as W

    hp = W.tiny_hparams()
    cfgm = {"upsample_rates": list(hp.up_rates)[:hp.n_ups], "resblock_dilation_sizes": [list(x)[:hp.n_resd] for x in hp.res_dilations][:hp.n_resk]}
    tens = W.make_synthetic_weights(hp, 77)
    dup = dict(tens)  # de-duplication: every LayerNorm gamma / beta identical, zero `post` convs, m == logs == 0
    for k in dup:
        if k.endswith(".gamma"):
            dup[k] = np.ones_like(dup[k])
        elif k.endswith(".beta") or ".post." in k or k in ("dp.flows.0.m", "dp.flows.0.logs"):
            dup[k] = np.zeros_like(dup[k])
    for case, t in (("distinct", tens), ("deduplicated", dup)):
        data = X.export_vits(G.ref_for(hp, t))
        g = oi.OnnxGraph(data)
        assert "enc_p.encoder.spk_emb_linear.weight" not in g.inits, "exporter behaviour changed: update the notes in OnnxGraph"
        if case == "deduplicated":
            assert len(g.alias) > 20
        hp2, got = oi.import_onnx(data, cfgm)
        assert W.pack_blob(hp2, got) == W.pack_blob(hp, t), case


@pytest.mark.skipif(not _reference_present(), reason="container-only: runs the reference's own ONNX export procedures")
def test_real_exports_of_the_multistream_model_and_of_bert_import_bit_exactly():
    """matcha/onnx/export.py's MatchaWithVocoder (reference MatchaTTS + in-tree HiFi-GAN V1, synthetic weights) and
    bert-export.py's BertModel -> real torch.onnx exports -> importers -> byte-identical STTSW001 / VITSW001 / BERTW001 blobs,
    every hyper-parameter (incl. n_timesteps, guidance scale, mel statistics read off the traced graph) recovered."""
    import sys

    import torch

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import onnx_export_ref as X

    BertConfig, BertModel = X.transformers_bert()
    from vosk_tts_amd import onnx_import as oi
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_bert as BW
    from vosk_tts_amd import weights_stts as S

    hp = BW.small_hparams(120, 128, 4)
    cfg = BertConfig(vocab_size=hp.vocab_size, hidden_size=hp.hidden, num_hidden_layers=hp.n_layers, num_attention_heads=hp.n_heads,
                     intermediate_size=hp.intermediate, max_position_embeddings=hp.max_position, type_vocab_size=hp.type_vocab,
                     layer_norm_eps=hp.ln_eps, hidden_act="gelu", attn_implementation="eager")
    net = BertModel(cfg).eval()
    tens = BW.make_synthetic_weights(hp, 31)
    sd = net.state_dict()
    with torch.no_grad():
        for k, v in tens.items():
            sd[k].copy_(torch.from_numpy(v))
    hp2, got = oi.import_bert_onnx(X.export_bert(net), {"n_heads": hp.n_heads})
    assert BW.pack_blob(hp2, got) == BW.pack_blob(hp, tens)

    import gen_golden_stts as G

    shp, snet, voc = G.build()  # reference MatchaTTS + HiFi-GAN V1 carrying the synthetic blobs' tensors
    voc.decode = voc.forward   # export.py:30 calls vocoder.decode(): Vocos' method; the in-tree generator is called directly (cli.py)
    data = X.export_stts(snet, voc, n_timesteps=3)
    hp3, t3, v3 = oi.import_stts_onnx(data)
    shp.n_timesteps = 3
    assert S.pack_blob(hp3, t3) == S.pack_blob(shp, S.make_synthetic_weights(shp, G.SEED))
    vhp = W.hifigan_v1_vocoder_hparams()
    assert W.pack_blob(*v3) == W.pack_blob(vhp, W.make_synthetic_weights(vhp, G.SEED))
    # a foreign embedded vocoder is reported, not guessed
    g = oi.OnnxGraph(data)
    foreign = {("vocoder.backbone." + k[len("vocoder."):] if k.startswith("vocoder.") else k): v for k, v in g.inits.items() if v.dtype == np.float32 and v.size > 1}
    with pytest.raises(NotImplementedError, match="Vocos"):
        oi.import_stts_onnx(oi.write_minimal_onnx(os.path.join(os.environ.get("TMPDIR", "/tmp"), "foreign_voc.onnx"), foreign))


def test_multistream_and_bert_files_written_by_the_minimal_writer_import(tmp_path):
    """portable twin of the container-only test above: initializer-only files under the exporter's names"""
    from vosk_tts_amd import onnx_import as oi
    from vosk_tts_amd import weights as W
    from vosk_tts_amd import weights_bert as BW
    from vosk_tts_amd import weights_stts as S
    from vosk_tts_amd.toymodel import multistream_dir_to_reference_layout, write_toy_multistream_model

    d = write_toy_multistream_model(str(tmp_path / "ms"), with_bert=True)
    blobs = {n: open(os.path.join(d, n), "rb").read() for n in ("model.sttsw", "vocoder.vitsw", "bert/model.bertw")}
    multistream_dir_to_reference_layout(d)
    assert not os.path.exists(os.path.join(d, "model.sttsw")) and os.path.exists(os.path.join(d, "bert", "model.onnx"))
    hp, t, voc = oi.import_stts_onnx(os.path.join(d, "model.onnx"))
    assert S.pack_blob(hp, t) == blobs["model.sttsw"] and W.pack_blob(*voc) == blobs["vocoder.vitsw"]
    bhp, bt = oi.import_bert_onnx(os.path.join(d, "bert", "model.onnx"))
    assert BW.pack_blob(bhp, bt) == blobs["bert/model.bertw"]
    with pytest.raises(NotImplementedError, match="multistream"):
        oi.import_stts_onnx(oi.write_minimal_onnx(str(tmp_path / "x.onnx"), {"enc_p.emb.weight": np.zeros((4, 4), np.float32)}))


def test_node_scope_candidates():
    """module path -> node-name scopes as the TorchScript exporter writes them: a numbered child of a ModuleList keeps the
    list's name ("/flows.0"), one of a Sequential sits inside the Sequential's own scope ("/bert_proj/bert_proj.1")"""
    from vosk_tts_amd.onnx_import import OnnxGraph

    assert OnnxGraph.scope("flow.flows.0.post") == "/flow/flows.0/post/"
    assert "/matcha/encoder/bert_proj/bert_proj.1/" in OnnxGraph.scopes("matcha.encoder.bert_proj.1")
    assert "/enc_p/encoder/norm_layers_1.3/" in OnnxGraph.scopes("enc_p.encoder.norm_layers_1.3")
    assert OnnxGraph.scopes("dp.pre") == ["/dp/pre/"]


def test_wire_reader_roundtrips_random_tensors():
    """property test of the hand-rolled protobuf reader against the independent minimal writer: random names, ranks, shapes
    (incl. zero-sized and scalar tensors), raw_data and float_data encodings"""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    from vosk_tts_amd import onnx_import as oi

    shapes = st.lists(st.integers(0, 5), min_size=0, max_size=4)
    names = st.text(alphabet="abcdefghijklmnopqrstuvwxyz._0123456789:", min_size=1, max_size=40)

    @settings(max_examples=60, deadline=None)
    @given(st.dictionaries(names, shapes, min_size=1, max_size=6), st.integers(0, 2**31 - 1))
    def check(spec, seed):
        import tempfile

        rng = np.random.default_rng(seed)
        tens = {k: rng.standard_normal(tuple(v)).astype(np.float32) for k, v in spec.items()}
        floaty = {k for i, k in enumerate(tens) if i % 2 and tens[k].size}
        with tempfile.TemporaryDirectory() as d:
            got = oi.read_initializers(oi.write_minimal_onnx(os.path.join(d, "m.onnx"), tens, use_float_data=floaty))
        assert set(got) == set(tens)
        for k in tens:
            assert got[k].shape == tens[k].shape and np.array_equal(got[k], tens[k]), k

    check()


def test_decoder_geometry_comes_from_the_graph_and_is_validated(tmp_path):
    """A runtime model's config.json has no "model" section: upsample rates must come from the ConvTranspose `strides`, ResBlock
    dilations from the Conv `dilations`, hop_length from their product; inconsistent geometry is rejected before a blob exists
    (engine.hip makes the same checks at vits_create: a zero rate would otherwise divide by zero, a wrong hop_length overrun)."""
    from vosk_tts_amd import onnx_import as O
    from vosk_tts_amd import weights as W

    hp = W.plain_hparams()  # plain HiFi-GAN generator: rates other than the in-repo default [4, 4]
    rates = [hp.up_rates[i] for i in range(hp.n_ups)]
    t = W.make_synthetic_weights(hp, 5)
    nodes = [(f"/dec/ups.{i}/ConvTranspose", "ConvTranspose", ["x", f"dec.ups.{i}.weight"], ["y"], {"strides": [r], "kernel_shape": [hp.up_kernels[i]]})
             for i, r in enumerate(rates)]
    nodes += [(f"/dec/resblocks.{j}/convs1.{d}/Conv", "Conv", ["x", f"dec.resblocks.{j}.convs1.{d}.weight"], ["y"], {"dilations": [hp.res_dilations[j][d]]})
              for j in range(hp.n_resk) for d in range(hp.n_resd)]
    p = O.write_minimal_onnx(str(tmp_path / "m.onnx"), t, nodes=nodes)
    got, _ = O.import_onnx(p)  # no config at all
    assert [got.up_rates[i] for i in range(got.n_ups)] == rates
    assert got.hop_length == int(np.prod(rates)) == hp.hop_length
    assert [[got.res_dilations[j][d] for d in range(got.n_resd)] for j in range(got.n_resk)] == \
           [[hp.res_dilations[j][d] for d in range(hp.n_resd)] for j in range(hp.n_resk)]
    with pytest.raises(ValueError, match="hop_length"):
        O.import_onnx(p, {"hop_length": 2 * hp.hop_length, "upsample_rates": rates})  # contradicts the graph
    bad = W.default_hparams()
    bad.up_rates[1] = 0
    with pytest.raises(ValueError, match="upsample rate"):
        W.pack_blob(bad, {})
    bad = W.default_hparams()
    bad.hop_length = 512
    with pytest.raises(ValueError, match="hop_length"):
        W.pack_blob(bad, {})


def test_bert_conditioned_vits_graph_surfaces_its_projection(tmp_path):
    """A BERT-conditioned VITS export (vosk_tts/synth.py:88-99) carries one extra projection; whatever the exporter called it,
    a [hidden, D] / [hidden, D, 1] weight and a [hidden] bias with "bert" in their names map onto enc_p.bert_proj and set
    bert_dim; an unidentifiable set of BERT-like tensors is reported with names and shapes."""
    from vosk_tts_amd import onnx_import as O
    from vosk_tts_amd import weights as W

    hp = W.tiny_hparams()
    hp.bert_dim = 32
    t = W.make_synthetic_weights(hp, 3)
    w, b = t.pop("enc_p.bert_proj.weight"), t.pop("enc_p.bert_proj.bias")
    g = dict(t)
    g["enc_p.bert_linear.weight"] = w[:, :, 0]  # nn.Linear layout
    g["enc_p.bert_linear.bias"] = b
    got_hp, got = O.import_onnx(O.write_minimal_onnx(str(tmp_path / "a.onnx"), g))
    assert got_hp.bert_dim == 32 and np.array_equal(got["enc_p.bert_proj.weight"], w) and np.array_equal(got["enc_p.bert_proj.bias"], b)
    assert any("bert_linear" in n for n in O.import_onnx.notes)
    g["enc_p.bert_gate.weight"] = w[:, :, 0]  # a second candidate: refuse to guess
    with pytest.raises(NotImplementedError, match="bert_gate"):
        O.import_onnx(O.write_minimal_onnx(str(tmp_path / "b.onnx"), g))
