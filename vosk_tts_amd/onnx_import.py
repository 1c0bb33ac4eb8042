"""Import the weights of a shipped `model.onnx`.

The reference runtime loads `model.onnx` with onnxruntime (vosk_tts/model.py:46); the graph was
written by `torch.onnx.export` from `SynthesizerTrn` with weight-norm removed from `dec` and `flow`
(training/vits2/onnx_export.py:47-104), so its *initializers* carry the parameter names of the
PyTorch module ("dec.conv_pre.weight", "enc_p.encoder.attn_layers.0.conv_q.bias", ...).  This
module reads those initializers with a minimal protobuf wire-format parser (no `onnx` package is
needed), infers the hyper-parameters from the tensor shapes and produces the engine's VITSW001 blob.

Limits (stated, not hidden): only graphs of the in-repo VITS2 family are recognised; tensors that
the exporter constant-folded under anonymous names ("onnx::Conv_123") cannot be matched and are
reported as missing; BERT-conditioned / multistream flavours (vosk_tts/synth.py:64-99) are rejected.
This path could not be validated against a real vosk model here (none is available offline); the
wire parser and the name/shape mapping are covered by tests/test_onnx_import.py with a synthetic file.

ONNX protobuf fields used (onnx.proto3): ModelProto.graph = 7; GraphProto.node = 1, .initializer = 5;
NodeProto.output = 2, .op_type = 4, .attribute = 5; AttributeProto.name = 1, .i = 3, .t = 5, .ints = 8;
TensorProto.dims = 1, .data_type = 2, .float_data = 4, .int64_data = 7, .name = 8, .raw_data = 9,
.double_data = 10, .data_location = 14.
"""
import struct

import numpy as np

from . import weights as W

# ---------------------------------------------------------------------------------------------------
# protobuf wire format
# ---------------------------------------------------------------------------------------------------


def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf):
    """yield (field_number, wire_type, value) for one message; value is int (varint / fixed) or memoryview (bytes)"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]; pos += ln
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        if pos > n:
            raise ValueError("truncated protobuf message")
        yield fno, wt, v


_DTYPES = {1: np.float32, 7: np.int64, 11: np.float64, 6: np.int32, 10: np.float16}


def _packed_varints(v):
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x)
    return out


def _tensor(buf):
    """TensorProto -> (name, ndarray or None)"""
    dims, dtype, name, raw, floats, int64s, doubles, external = [], 1, "", None, [], [], [], False
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims.extend(_packed_varints(v) if wt == 2 else [v])
        elif fno == 2:
            dtype = v
        elif fno == 4:
            floats.append(np.frombuffer(bytes(v), "<f4") if wt == 2 else np.frombuffer(v, "<f4"))
        elif fno == 7:
            int64s.extend(_packed_varints(v) if wt == 2 else [v])
        elif fno == 8:
            name = bytes(v).decode("utf-8", "replace")
        elif fno == 9:
            raw = bytes(v)
        elif fno == 10:
            doubles.append(np.frombuffer(bytes(v), "<f8") if wt == 2 else np.frombuffer(v, "<f8"))
        elif fno == 14 and v == 1:
            external = True
    if external:
        return name, None
    np_dt = _DTYPES.get(dtype)
    if np_dt is None:
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_dt).newbyteorder("<"))
    elif floats:
        arr = np.concatenate(floats)
    elif doubles:
        arr = np.concatenate(doubles)
    elif int64s:
        arr = np.array([x - (1 << 64) if x >= (1 << 63) else x for x in int64s], dtype=np.int64)
    else:
        arr = np.zeros(0, np_dt)
    shape = tuple(int(d) for d in dims)
    if int(np.prod(shape, dtype=np.int64)) != arr.size:
        return name, None
    return name, arr.reshape(shape)


class OnnxGraph:
    """Initializers + nodes of one ONNX file, with the exporter's renamings undone.

    What `torch.onnx.export` (TorchScript exporter, the one training/vits2/onnx_export.py:94-110 and
    matcha/onnx/export.py:96-117 call) does to parameters, as observed on real exports of the reference modules
    (oracle/gen_onnx_fixtures.py):
      * a parameter normally becomes an initializer under its state_dict name;
      * byte-identical parameters are de-duplicated: one initializer survives and every other name is the OUTPUT of
        an `Identity` node fed by it (e.g. untrained LayerNorm gammas, zero-initialised `post` convs);
      * an `nn.Linear` weight is constant-folded into its transpose under an anonymous name ("onnx::MatMul_123") that
        feeds the node "/<module/path>/MatMul"; a bias may likewise sit behind "/<module/path>/Add";
      * other folded expressions keep no name at all: ElementwiseAffine's `exp(-logs)` leaves "onnx::Exp_N" = -logs
        feeding "/dp/flows.0/Exp";
      * node names carry the module path: "a.b.0.c" -> "/a/b.0/c/<Op>[_k]".
    `param(name)` resolves a state_dict name through all of these."""

    def __init__(self, path_or_bytes):
        data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray, memoryview)) else open(path_or_bytes, "rb").read()
        self.inits = {}
        self.nodes = []  # (name, op_type, inputs, outputs)
        self.attrs = {}  # node name -> {attribute name: [ints]}  (strides / dilations / kernel_shape of the conv nodes)
        self.alias = {}  # Identity output -> input
        for fno, wt, v in _fields(memoryview(data)):
            if fno != 7 or wt != 2:  # ModelProto.graph
                continue
            for gno, gwt, gv in _fields(v):
                if gno == 5 and gwt == 2:  # initializer
                    name, arr = _tensor(gv)
                    if arr is not None:
                        self.inits[name] = arr
                elif gno == 1 and gwt == 2:  # node
                    ins, outs, name, op, tens, attrs = [], [], "", "", None, {}
                    for nno, nwt, nv in _fields(gv):
                        if nno == 1:
                            ins.append(bytes(nv).decode("utf-8", "replace"))
                        elif nno == 2:
                            outs.append(bytes(nv).decode("utf-8", "replace"))
                        elif nno == 3:
                            name = bytes(nv).decode("utf-8", "replace")
                        elif nno == 4:
                            op = bytes(nv).decode("utf-8", "replace")
                        elif nno == 5 and nwt == 2:
                            aname, ints = "", []
                            for ano, awt, av in _fields(nv):
                                if ano == 5 and awt == 2:
                                    tens = _tensor(av)[1]
                                elif ano == 1 and awt == 2:
                                    aname = bytes(av).decode("utf-8", "replace")
                                elif ano == 8:  # repeated int64 ints (packed or one varint per element)
                                    ints.extend(_packed_varints(av) if awt == 2 else [int(av)])
                                elif ano == 3 and awt == 0:  # single int
                                    ints.append(int(av))
                            if aname and ints:
                                attrs[aname] = ints
                    self.nodes.append((name, op, ins, outs))
                    if name and attrs:
                        self.attrs[name] = attrs
                    if op == "Constant" and tens is not None and outs:  # Constant tensors count as initializers
                        self.inits.setdefault(outs[0], tens)
                    elif op == "Identity" and len(ins) == 1 and len(outs) == 1:
                        self.alias[outs[0]] = ins[0]
        self._by_name = {n[0]: n for n in self.nodes if n[0]}

    def tensors(self):
        """name -> ndarray: the initializers plus every de-duplicated name (Identity outputs)."""
        out = dict(self.inits)
        for name in self.alias:
            a = self._follow(name)
            if a is not None:
                out.setdefault(name, a)
        return out

    def _follow(self, name):
        for _ in range(8):
            if name in self.inits:
                return self.inits[name]
            if name not in self.alias:
                return None
            name = self.alias[name]
        return None

    @staticmethod
    def scope(module_path):
        """state_dict module path -> node-name scope: "flow.flows.0.post" -> "/flow/flows.0/post/"."""
        return OnnxGraph.scopes(module_path)[0]

    @staticmethod
    def scopes(module_path):
        """candidate node-name scopes of a module path.  A numbered child of a ModuleList (never called itself) appears as
        "/flows.0", one of a Sequential (called, so it has a scope of its own) as "/bert_proj/bert_proj.1"."""
        outs = [""]
        prev = ""
        for part in module_path.split("."):
            if part.isdigit():
                outs = [o + "." + part for o in outs] + [o + "/" + prev + "." + part for o in outs]
            else:
                outs = [o + "/" + part for o in outs]
            prev = part
        return [o + "/" for o in outs]

    def _node_init(self, node_name, index=None):
        n = self._by_name.get(node_name)
        if n is None:
            return None
        cands = [n[2][index]] if index is not None and index < len(n[2]) else n[2]
        for i in cands:
            a = self._follow(i)
            if a is not None:
                return a
        return None

    def param(self, name):
        """ndarray for a state_dict name, or None."""
        a = self._follow(name)
        if a is not None:
            return a
        mod, _, leaf = name.rpartition(".")
        for sc in self.scopes(mod):
            a = self._param_in_scope(sc, leaf)
            if a is not None:
                return a
        return None

    def _param_in_scope(self, sc, leaf):
        if leaf == "weight":
            a = self._node_init(sc + "MatMul")  # nn.Linear: folded transpose
            if a is not None and a.ndim == 2:
                return np.ascontiguousarray(a.T)
            for op in ("Conv", "ConvTranspose"):
                a = self._node_init(sc + op, 1)
                if a is not None:
                    return a
        elif leaf == "bias":
            for op in ("Conv", "ConvTranspose"):
                a = self._node_init(sc + op, 2)
                if a is not None:
                    return a
            a = self._node_init(sc + "Add")
            if a is not None:
                return a
        elif leaf == "logs":  # ElementwiseAffine reverse (modules.py): x * exp(-logs) -> the folded constant is -logs
            a = self._node_init(sc + "Exp")
            if a is not None:
                return -np.asarray(a) + 0.0  # + 0.0: -(+0) would give -0
        return None

    def anonymous(self, prefix, shape):
        """the unique anonymous initializer "onnx::<prefix>_N" of this shape (exporters that do not name nodes), or None."""
        c = [a for k, a in self.inits.items() if k.startswith("onnx::" + prefix) and tuple(a.shape) == tuple(shape)]
        return c[0] if len(c) == 1 else None


def read_initializers(path_or_bytes):
    """name -> ndarray for every initializer, Constant node output and de-duplicated parameter name of an ONNX file."""
    return OnnxGraph(path_or_bytes).tensors()


# ---------------------------------------------------------------------------------------------------
# graph family recognition
# ---------------------------------------------------------------------------------------------------


def infer_hparams(t):
    """Hyper-parameters from tensor shapes (training/vits2/models.py:1503-1630 constructor wiring)."""
    def need(name):
        if name not in t:
            raise KeyError(name)
        return t[name]

    try:
        emb = need("enc_p.emb.weight")
        hp = W.default_hparams(n_vocab=emb.shape[0])
        hp.hidden_channels = emb.shape[1]
        f1 = need("enc_p.encoder.ffn_layers.0.conv_1.weight")
        hp.filter_channels, hp.kernel_size = f1.shape[0], f1.shape[2]
        hp.n_layers = sum(1 for k in t if k.startswith("enc_p.encoder.attn_layers.") and k.endswith(".conv_q.weight"))
        rel = need("enc_p.encoder.attn_layers.0.emb_rel_k")
        hp.window_size = (rel.shape[1] - 1) // 2
        hp.n_heads = hp.hidden_channels // rel.shape[2]
        hp.inter_channels = need("enc_p.proj.weight").shape[0] // 2
        if "emb_g.weight" in t:
            hp.n_speakers, hp.gin_channels = t["emb_g.weight"].shape
        else:
            hp.n_speakers, hp.gin_channels = 0, 0
        # the Linear's weight is folded into an anonymous transposed MatMul constant by the exporter; its bias keeps the name
        hp.enc_cond_layer = 2 if ("enc_p.encoder.spk_emb_linear.bias" in t or "enc_p.encoder.spk_emb_linear.weight" in t) else -1
        hp.dp_filter_channels = need("dp.pre.weight").shape[0]
        hp.dp_kernel_size = need("dp.convs.convs_sep.0.weight").shape[2]
        hp.dp_dds_layers = sum(1 for k in t if k.startswith("dp.convs.convs_sep.") and k.endswith(".weight"))
        cf = sorted({int(k.split(".")[2]) for k in t if k.startswith("dp.flows.") and k.endswith(".proj.weight")})
        hp.dp_n_flows = (max(cf) + 1) // 2 if cf else 4
        hp.dp_num_bins = (need(f"dp.flows.{max(cf)}.proj.weight").shape[0] + 1) // 3
        fl = sorted({int(k.split(".")[2]) for k in t if k.startswith("flow.flows.") and k.endswith(".pre.weight")})
        hp.flow_n_flows = len(fl)
        hp.flow_wn_layers = sum(1 for k in t if k.startswith("flow.flows.0.enc.in_layers.") and k.endswith(".weight"))
        hp.flow_kernel_size = need("flow.flows.0.enc.in_layers.0.weight").shape[2]
        hp.dec_initial_channel = need("dec.conv_pre.weight").shape[0]
        ups = sorted({int(k.split(".")[2]) for k in t if k.startswith("dec.ups.") and k.endswith(".weight")})
        hp.n_ups = len(ups)
        for i in ups:
            hp.up_kernels[i] = t[f"dec.ups.{i}.weight"].shape[2]
        n_rb = len({int(k.split(".")[2]) for k in t if k.startswith("dec.resblocks.")})
        hp.n_resk = n_rb // max(hp.n_ups, 1)
        for j in range(hp.n_resk):
            hp.res_kernels[j] = t[f"dec.resblocks.{j}.convs1.0.weight"].shape[2]
        hp.n_resd = sum(1 for k in t if k.startswith("dec.resblocks.0.convs1.") and k.endswith(".weight"))
        if "dec.subband_conv_post.weight" in t:
            hp.dec_type = 0
            post = t["dec.subband_conv_post.weight"].shape[0]
            hp.istft_n_fft = post // hp.subbands - 2
        elif "dec.conv_post.weight" in t:
            hp.dec_type = 1
        else:
            raise KeyError("dec.subband_conv_post.weight / dec.conv_post.weight")
    except KeyError as e:
        raise NotImplementedError(
            f"not an in-repo VITS2 graph: initializer {e.args[0]!r} not found (BERT / multistream flavours and graphs whose "
            "parameters were constant-folded under anonymous names are outside this importer, SURVEY.md §8f)") from None
    return hp


def import_onnx(path_or_bytes, config=None):  # noqa: C901
    """-> (HParams, {name: float32 ndarray}) restricted to the tensors the engine needs.

    `config`: optional dict with the values that shapes cannot reveal — "upsample_rates",
    "resblock_dilation_sizes", "gen_istft_hop_size", "subbands", "sampling_rate", "hop_length"
    (keys of training/vits2/configs/*.json "model"/"data")."""
    g = OnnxGraph(path_or_bytes)
    t = g.tensors()
    hp = infer_hparams(t)
    config = config or {}
    # BERT-conditioned flavours (vosk_tts/synth.py:88-99): their text encoder is not in the reference tree, so the extra tensors are
    # found by name and shape -- a [hidden, D(, 1)] weight with "bert" in its name and a [hidden] bias next to it -- and mapped onto
    # the engine's enc_p.bert_proj (1x1 projection of the bert feed added to the scaled embedding).  Anything else BERT-like is
    # reported with names and shapes instead of being guessed at.
    bertish = {k: a for k, a in t.items() if "bert" in k.lower() and not k.startswith("enc_p.bert_proj.")}
    if "enc_p.bert_proj.weight" in t:
        hp.bert_dim = int(t["enc_p.bert_proj.weight"].shape[1])
    elif bertish:
        H = hp.hidden_channels
        ws = [(k, a) for k, a in bertish.items() if a.ndim in (2, 3) and a.shape[0] == H and a.shape[1] >= 16 and (a.ndim == 2 or a.shape[2] == 1)]
        bs = [(k, a) for k, a in bertish.items() if a.ndim == 1 and a.shape[0] == H]
        if len(ws) == 1 and len(bs) == 1:
            (wk, w), (bk, b) = ws[0], bs[0]
            hp.bert_dim = int(w.shape[1])
            t = dict(t)
            t["enc_p.bert_proj.weight"] = np.ascontiguousarray(np.asarray(w, np.float32).reshape(H, hp.bert_dim, 1))
            t["enc_p.bert_proj.bias"] = np.asarray(b, np.float32)
            import_onnx.notes = [f"bert projection: {wk} {tuple(w.shape)} + {bk} -> enc_p.bert_proj (bert_dim {hp.bert_dim})"]
        else:
            raise NotImplementedError("BERT-conditioned graph whose projection could not be identified; BERT-like tensors: "
                                      + ", ".join(f"{k} {tuple(a.shape)}" for k, a in sorted(bertish.items())[:12]))
    # Geometry that tensor shapes cannot reveal comes from the graph itself: the `strides` attribute of every
    # /dec/ups.N/ConvTranspose node, the `dilations` of the ResBlock convs, the stride of the iSTFT's ConvTranspose.
    # (A runtime vosk model's config.json normally has no "model"/"data" sections; config values, when present, win.)
    for i in range(hp.n_ups):
        st = g.attrs.get(f"/dec/ups.{i}/ConvTranspose", {}).get("strides")
        if st:
            hp.up_rates[i] = int(st[0])
        elif hp.up_rates[i] <= 0:  # no attribute and no in-repo default for this stage: HiFi-GAN convention kernel = 2 * rate
            hp.up_rates[i] = max(1, hp.up_kernels[i] // 2)
    for i in range(hp.n_ups, len(hp.up_rates)):
        hp.up_rates[i] = 0
    for j in range(hp.n_resk):
        for d in range(hp.n_resd):
            dl = g.attrs.get(f"/dec/resblocks.{j}/convs1.{d}/Conv", {}).get("dilations")
            if dl:
                hp.res_dilations[j][d] = int(dl[0])
    if hp.dec_type == 0:
        for nm, at in g.attrs.items():
            if nm.startswith("/dec/") and "stft" in nm.lower() and nm.endswith("ConvTranspose") and at.get("strides"):
                hp.istft_hop = int(at["strides"][0])
    for i, r in enumerate(config.get("upsample_rates", [])):
        hp.up_rates[i] = int(r)
    for j, dl in enumerate(config.get("resblock_dilation_sizes", [])):
        for d, v in enumerate(dl):
            hp.res_dilations[j][d] = int(v)
    for key, field in (("gen_istft_hop_size", "istft_hop"), ("subbands", "subbands"), ("sampling_rate", "sampling_rate"),
                       ("hop_length", "hop_length")):
        if key in config:
            setattr(hp, field, int(config[key]))
    rate = int(np.prod([hp.up_rates[i] for i in range(hp.n_ups)])) * (hp.istft_hop * hp.subbands if hp.dec_type == 0 else 1)
    if "hop_length" not in config:
        hp.hop_length = rate  # samples per frame are a property of the decoder, not an independent setting
    W.validate_hparams(hp)
    tensors, missing, bad = {}, [], []
    for name, shape, _kind, _fan, _gain in W.tensor_specs(hp):
        a = t.get(name)
        if a is None:
            a = g.param(name)
        if a is None and name.endswith(".weight") and len(shape) == 2:  # Linear weight of an exporter without node names
            a = g.anonymous("MatMul", shape[::-1])
            a = None if a is None else np.ascontiguousarray(a.T)
        if a is None and name.endswith(".logs"):
            a = g.anonymous("Exp", shape)
            a = None if a is None else -np.asarray(a) + 0.0
        if a is None:
            missing.append(name)
            continue
        a = np.asarray(a, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            bad.append(f"{name}: {tuple(a.shape)} != {tuple(shape)}")
            continue
        tensors[name] = np.ascontiguousarray(a)
    if missing or bad:
        raise NotImplementedError("model.onnx does not match the VITS2 tensor inventory: "
                                  f"{len(missing)} missing (e.g. {missing[:3]}), {len(bad)} with unexpected shapes (e.g. {bad[:2]})")
    return hp, tensors


def convert(onnx_path, blob_path, config=None):
    hp, tensors = import_onnx(onnx_path, config)
    W.save_blob(blob_path, hp, tensors)
    return hp


# ---------------------------------------------------------------------------------------------------
# StableTTS / Matcha "multistream" graphs (training/stabletts/matcha/onnx/export.py) and bert/model.onnx
# ---------------------------------------------------------------------------------------------------


def _collect(g, prefix, specs, what):
    """tensors of `specs` from graph `g` (state_dict names under `prefix`), loud about anything missing or misshapen"""
    t = g.tensors()
    out, missing, bad = {}, [], []
    for name, shape, *_ in specs:
        a = t.get(prefix + name)
        if a is None:
            a = g.param(prefix + name)
        if a is None and name.endswith(".weight") and len(shape) == 2:
            a = g.anonymous("MatMul", shape[::-1])
            a = None if a is None else np.ascontiguousarray(a.T)
        if a is None:
            missing.append(name)
            continue
        a = np.asarray(a, dtype=np.float32)
        if tuple(a.shape) != tuple(shape):
            bad.append(f"{name}: {tuple(a.shape)} != {tuple(shape)}")
            continue
        out[name] = np.ascontiguousarray(a)
    if missing or bad:
        raise NotImplementedError(f"model.onnx does not match the {what} tensor inventory: {len(missing)} missing (e.g. {missing[:3]}), "
                                  f"{len(bad)} with unexpected shapes (e.g. {bad[:2]})")
    return out


def _count(prefix_re, t):
    import re

    r = re.compile(prefix_re)
    return len({m.group(1) for k in t for m in [r.match(k)] if m})


def import_stts_onnx(path_or_bytes, config=None):
    """A multistream `model.onnx` -> (SttsHParams, acoustic tensors, vocoder) where vocoder is (HParams, tensors) of a
    vocoder-only VITSW001 blob, or None when the embedded vocoder is not the in-tree HiFi-GAN generator.

    The exported module is MatchaWithVocoder (export.py:21-32): parameters sit under "matcha." and "vocoder."; a mel-only
    export (vocoder None, :56-58) has no prefix.  Sizes come from tensor shapes; what shapes cannot tell is read off the graph:
    n_timesteps and guidance_scale from the traced Euler loop (estimator calls, the scalar of the guidance blend), mel_std / mel_mean =
    the scalars of the final `mel * std + mean` (matcha/utils/model.py denormalize) at "/matcha/Mul_k" -> "/matcha/Add".
    `config` may override: n_timesteps, guidance_scale, mel_mean, mel_std, enc_heads, dec_heads, hop_length, sampling_rate,
    upsample_rates (vocoder).  Reference's export calls `vocoder.decode(mel)` - a method the third-party Vocos has and the
    in-tree HiFi-GAN `Generator` has not (matcha/cli.py:72-98); a Vocos / BigVGAN vocoder is reported, not imported."""
    from . import weights_stts as S

    config = config or {}
    g = OnnxGraph(path_or_bytes)
    t = g.tensors()
    pre = "matcha." if any(k.startswith("matcha.") for k in t) else ""
    root = "/matcha" if pre else ""

    def need(name):
        if pre + name not in t:
            raise NotImplementedError(f"not a StableTTS / Matcha multistream graph: initializer {pre + name!r} not found")
        return t[pre + name]

    hp = S.default_hparams()
    emb, punc = need("encoder.emb.weight"), need("encoder.punc_emb.weight")
    hp.n_vocab, hp.emb_dim, hp.punc_dim = emb.shape[0], emb.shape[1], punc.shape[1]
    hp.bert_proj_dim = need("encoder.bert_proj.1.bias").shape[0]
    e0 = "encoder.dp_encoder.encoder.0"
    hp.enc_hidden = need(e0 + ".attn.conv_q.weight").shape[0]
    c1 = need(e0 + ".mlp.conv_1.weight")
    hp.enc_filter, hp.enc_kernel = c1.shape[0], c1.shape[2]
    hp.enc_layers = _count(r"(?:matcha\.)?encoder\.dp_encoder\.encoder\.(\d+)\.attn\.conv_q\.weight", t)
    hp.spk_emb_dim = need(e0 + ".adaLN_modulation.0.weight").shape[1]
    hp.dp_out = need("encoder.dp_encoder.proj.weight").shape[0]
    hp.bert_dim = 768
    w = g.param(pre + "encoder.bert_proj.1.weight")
    if w is not None:
        hp.bert_dim = w.shape[1]
    hp.n_spks = t[pre + "spk_emb.weight"].shape[0] if pre + "spk_emb.weight" in t else 1
    d = "decoder.estimator"
    hp.dec_hidden = need(d + ".blocks.0.block.attn.conv_q.weight").shape[0]
    c1 = need(d + ".blocks.0.block.mlp.conv_1.weight")
    hp.dec_filter, hp.dec_kernel = c1.shape[0], c1.shape[2]
    hp.dec_layers = _count(r"(?:matcha\.)?decoder\.estimator\.blocks\.(\d+)\.block\.attn\.conv_q\.weight", t)
    hp.n_feats = need(d + ".final_proj.weight").shape[0]
    import re

    # every traced estimator call repeats the module's nodes under "<module>_k"; with classifier-free guidance there are two
    # calls per Euler step (flow_matching.py:177-189) and a `dphi + g * (dphi - dphi_avg)`: Sub -> Mul(scalar g) -> Add
    calls = {n[0] for n in g.nodes if re.fullmatch(re.escape(root) + r"/decoder/estimator/in_proj(_\d+)?/Conv", n[0])}
    subs = {n[3][0] for n in g.nodes if n[1] == "Sub" and re.fullmatch(re.escape(root) + r"/decoder/Sub(_\d+)?", n[0]) and n[3]}
    gs = []
    for n in g.nodes:
        if n[1] == "Mul" and re.fullmatch(re.escape(root) + r"/decoder/Mul(_\d+)?", n[0]) and any(i in subs for i in n[2]):
            gs += [float(g._follow(i).reshape(-1)[0]) for i in n[2] if g._follow(i) is not None and g._follow(i).size == 1]
    if gs and max(gs) - min(gs) < 1e-7:
        hp.guidance_scale = gs[0]
        hp.n_timesteps = len(gs)
    elif calls:
        hp.guidance_scale = 0.0
        hp.n_timesteps = len(calls)
    add = g._by_name.get(root + "/Add")  # denormalize: the last arithmetic of synthesise (matcha_tts.py:205)
    if add is not None:
        mean = [g._follow(i) for i in add[2] if g._follow(i) is not None and g._follow(i).size == 1]
        prod = [n for n in g.nodes if n[3] and n[3][0] in add[2] and n[1] == "Mul"]
        std = [g._follow(i) for n in prod for i in n[2] if g._follow(i) is not None and g._follow(i).size == 1]
        if len(mean) == 1 and len(std) == 1:
            hp.mel_mean, hp.mel_std = float(mean[0].reshape(-1)[0]), float(std[0].reshape(-1)[0])
    for key in ("n_timesteps", "enc_heads", "dec_heads", "hop_length", "sampling_rate"):
        if key in config:
            setattr(hp, key, int(config[key]))
    for key in ("guidance_scale", "mel_mean", "mel_std"):
        if key in config:
            setattr(hp, key, float(config[key]))
    specs = S.tensor_specs(hp)
    if hp.n_spks <= 1 and pre + "spk_emb.weight" not in t:  # single-speaker checkpoints have no speaker tables (matcha_tts.py:62-64)
        specs = [sp for sp in specs if sp[0] not in ("spk_emb.weight", "dur_spk_emb.weight")]
    tensors = _collect(g, pre, specs, "StableTTS")
    for nm in ("spk_emb.weight", "dur_spk_emb.weight"):
        tensors.setdefault(nm, np.zeros((1, hp.spk_emb_dim), np.float32))
    tensors = {n: tensors[n] for n, *_ in S.tensor_specs(hp)}
    # ---- embedded vocoder
    vocoder = None
    vnames = [k for k in t if k.startswith("vocoder.")]
    if vnames:
        if "vocoder.conv_pre.weight" in t and "vocoder.ups.0.weight" in t and "vocoder.conv_post.weight" in t:
            vocoder = _import_hifigan(g, t, "vocoder.", config)
        else:
            fam = "Vocos" if any("backbone" in k or "head.out" in k for k in vnames) else "unknown"
            raise NotImplementedError(f"embedded vocoder of family {fam!r} ({vnames[0]} ...): only the in-tree HiFi-GAN generator "
                                      "(matcha/hifigan/models.py) is built; pass a mel-only export or see DESIGN.md section 8")
    return hp, tensors, vocoder


def _import_hifigan(g, t, pre, config):
    """HiFi-GAN generator (matcha/hifigan/models.py:148-199, weight norm removed) -> vocoder-only VITSW001 tensors ("dec." names)"""
    hp = W.hifigan_v1_vocoder_hparams()
    cp = t[pre + "conv_pre.weight"]
    hp.dec_initial_channel, hp.inter_channels = cp.shape[0], cp.shape[1]
    ups = sorted({int(k.split(".")[2]) for k in t if k.startswith(pre + "ups.") and k.endswith(".weight")})
    hp.n_ups = len(ups)
    rates = config.get("upsample_rates")
    for i in ups:
        k = t[f"{pre}ups.{i}.weight"].shape[2]
        hp.up_kernels[i] = k
        hp.up_rates[i] = int(rates[i]) if rates else k // 2  # every HiFi-GAN config of the reference uses kernel = 2 x rate (hifigan/config.py)
    n_rb = len({int(k.split(".")[2]) for k in t if k.startswith(pre + "resblocks.")})
    hp.n_resk = n_rb // max(hp.n_ups, 1)
    for j in range(hp.n_resk):
        hp.res_kernels[j] = t[f"{pre}resblocks.{j}.convs1.0.weight"].shape[2]
    hp.n_resd = sum(1 for k in t if k.startswith(pre + "resblocks.0.convs1.") and k.endswith(".weight"))
    for j, dl in enumerate(config.get("resblock_dilation_sizes", [])):
        for d, v in enumerate(dl):
            hp.res_dilations[j][d] = int(v)
    specs = [(n[len("dec."):],) + tuple(rest) for n, *rest in W.tensor_specs(hp)]
    tensors = _collect(g, pre, specs, "HiFi-GAN vocoder")
    return hp, {"dec." + k: v for k, v in tensors.items()}


def import_bert_onnx(path_or_bytes, config=None):
    """`bert/model.onnx` (matcha/onnx/bert-export.py: BertModel, output hidden_states[-3]) -> (BertHParams, tensors).
    Constant folding prunes the two unused top layers and the pooler, so the layer count of the file IS out_layers; Linear
    weights are folded transposes behind "/.../MatMul" nodes, LayerNorm parameters keep their names.  `config`: n_heads
    (default hidden // 64, BERT-base), ln_eps (1e-12)."""
    from . import weights_bert as B

    config = config or {}
    g = OnnxGraph(path_or_bytes)
    t = g.tensors()
    keys = [k for k in t if k.endswith("embeddings.word_embeddings.weight")]
    if not keys:
        raise NotImplementedError("not a BertModel export: no embeddings.word_embeddings.weight initializer")
    pre = keys[0][: -len("embeddings.word_embeddings.weight")]
    hp = B.base_hparams()
    we = t[keys[0]]
    hp.vocab_size, hp.hidden = we.shape
    hp.max_position = t[pre + "embeddings.position_embeddings.weight"].shape[0]
    hp.type_vocab = t[pre + "embeddings.token_type_embeddings.weight"].shape[0]
    hp.out_layers = _count(re_escape(pre) + r"encoder\.layer\.(\d+)\.attention\.output\.LayerNorm\.weight", t)
    hp.n_layers = hp.out_layers + 2
    inter = g.param(pre + "encoder.layer.0.intermediate.dense.weight")
    if inter is None:
        inter = g.anonymous("MatMul", (hp.hidden, 4 * hp.hidden))
        inter = None if inter is None else inter.T
    if inter is None:
        raise NotImplementedError("BertModel export: intermediate.dense weight not found")
    hp.intermediate = inter.shape[0]
    hp.n_heads = int(config.get("n_heads", max(hp.hidden // 64, 1)))
    hp.ln_eps = float(config.get("ln_eps", 1e-12))
    return hp, _collect(g, pre, B.tensor_specs(hp), "BertModel")


def re_escape(s):
    import re

    return re.escape(s)


# tiny writer used by the tests (and handy for fixtures): the inverse of read_initializers for float32 tensors


def _enc_varint(x):
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        out.append(b | (0x80 if x else 0))
        if not x:
            return bytes(out)


def _enc_field(fno, wt, payload):
    if wt == 0:
        return _enc_varint(fno << 3) + _enc_varint(payload)
    return _enc_varint(fno << 3 | 2) + _enc_varint(len(payload)) + payload


def write_minimal_onnx(path, tensors, use_float_data=(), nodes=()):
    """ModelProto{ir_version, graph{node..., initializer...}} with raw_data (or float_data for names in use_float_data);
    nodes: (name, op_type, inputs, outputs[, {attribute: [ints]}]) tuples."""
    graph = bytearray()
    for node in nodes:
        name, op, ins, outs = node[:4]
        attrs = node[4] if len(node) > 4 else {}
        ab = b""
        for an, ints in attrs.items():  # AttributeProto{name = 1, ints = 8 (one varint per element), type = 20 (INTS = 7)}
            ab += _enc_field(5, 2, _enc_field(1, 2, an.encode()) + b"".join(_enc_field(8, 0, int(v)) for v in ints) + _enc_field(20, 0, 7))
        graph += _enc_field(1, 2, b"".join(_enc_field(1, 2, i.encode()) for i in ins) + b"".join(_enc_field(2, 2, o.encode()) for o in outs)
                            + _enc_field(3, 2, name.encode()) + _enc_field(4, 2, op.encode()) + ab)
    for name, a in tensors.items():
        a = np.asarray(a, dtype="<f4")  # (ascontiguousarray would turn a 0-d tensor into [1])
        tp = b"".join(_enc_field(1, 0, int(d)) for d in a.shape) + _enc_field(2, 0, 1) + _enc_field(8, 2, name.encode())
        if name in use_float_data:
            tp += _enc_field(4, 2, a.tobytes())
        else:
            tp += _enc_field(9, 2, a.tobytes())
        graph += _enc_field(5, 2, tp)
    model = _enc_field(1, 0, 8) + _enc_field(7, 2, bytes(graph))
    with open(path, "wb") as f:
        f.write(model)
    return path


if __name__ == "__main__":
    import json
    import sys

    cfg = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
    if cfg and "model" in cfg:
        cfg = dict(cfg["model"], **cfg.get("data", {}))
    h = convert(sys.argv[1], sys.argv[2], cfg)
    print(f"wrote {sys.argv[2]}: hidden {h.hidden_channels}, {h.n_layers} layers, {h.n_speakers} speakers")
