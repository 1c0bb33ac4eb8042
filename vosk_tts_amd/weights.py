"""Weight container ("VITSW001" blob), hyper-parameters and the build-owned
deterministic synthetic weight generator.

The real vosk-model-tts-ru-0.9-multi weights are not available offline
(SURVEY.md §0.3), so parity and benchmarks run on seeded synthetic weights with
the exact tensor names/shapes of the reference's state_dict after
remove_weight_norm (training/vits2/onnx_export.py:77-80; names enumerated from
SynthesizerTrn, training/vits2/models.py:1503-1630).  The same numpy generator
runs in the build container (where the values are also pushed into the imported
reference via load_state_dict to produce tests/golden/) and on the GPU box.

Blob layout: see include/vits_mi355.h.
"""
import ctypes
import struct
import zlib

import numpy as np

VITS_ABI_VERSION = 1
MAX_UPS = 4
MAX_RESK = 4
MAX_RESD = 4
MAGIC = b"VITSW001"


class HParams(ctypes.Structure):
    """ctypes mirror of `struct vits_hparams` (include/vits_mi355.h)."""

    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("n_vocab", ctypes.c_int32),
        ("hidden_channels", ctypes.c_int32),
        ("inter_channels", ctypes.c_int32),
        ("filter_channels", ctypes.c_int32),
        ("n_heads", ctypes.c_int32),
        ("n_layers", ctypes.c_int32),
        ("kernel_size", ctypes.c_int32),
        ("window_size", ctypes.c_int32),
        ("gin_channels", ctypes.c_int32),
        ("n_speakers", ctypes.c_int32),
        ("enc_cond_layer", ctypes.c_int32),
        ("dp_filter_channels", ctypes.c_int32),
        ("dp_kernel_size", ctypes.c_int32),
        ("dp_n_flows", ctypes.c_int32),
        ("dp_num_bins", ctypes.c_int32),
        ("dp_dds_layers", ctypes.c_int32),
        ("flow_n_flows", ctypes.c_int32),
        ("flow_wn_layers", ctypes.c_int32),
        ("flow_kernel_size", ctypes.c_int32),
        ("flow_dilation_rate", ctypes.c_int32),
        ("dec_type", ctypes.c_int32),
        ("dec_initial_channel", ctypes.c_int32),
        ("n_ups", ctypes.c_int32),
        ("up_rates", ctypes.c_int32 * MAX_UPS),
        ("up_kernels", ctypes.c_int32 * MAX_UPS),
        ("n_resk", ctypes.c_int32),
        ("res_kernels", ctypes.c_int32 * MAX_RESK),
        ("n_resd", ctypes.c_int32),
        ("res_dilations", (ctypes.c_int32 * MAX_RESD) * MAX_RESK),
        ("subbands", ctypes.c_int32),
        ("istft_n_fft", ctypes.c_int32),
        ("istft_hop", ctypes.c_int32),
        ("pqmf_taps", ctypes.c_int32),
        ("pqmf_cutoff", ctypes.c_float),
        ("pqmf_beta", ctypes.c_float),
        ("dp_tail_bound", ctypes.c_float),
        ("sampling_rate", ctypes.c_int32),
        ("hop_length", ctypes.c_int32),
        ("bert_dim", ctypes.c_int32),
        ("conv_precision", ctypes.c_int32),  # 0 fp32 (default), 1 split-bf16 decoder ResBlock convs at batch size ("bf16x3")
        ("reserved", ctypes.c_int32 * 6),
    ]

    def total_upsample(self):
        u = 1
        for i in range(self.n_ups):
            u *= self.up_rates[i]
        return u


class BlobEntry(ctypes.Structure):
    _fields_ = [
        ("name", ctypes.c_char * 96),
        ("ndim", ctypes.c_uint32),
        ("dims", ctypes.c_uint32 * 4),
        ("pad_", ctypes.c_uint32),
        ("offset", ctypes.c_uint64),
        ("nelem", ctypes.c_uint64),
    ]


def default_hparams(n_vocab=62):
    """The in-repo MB-iSTFT-VITS2 config
    (training/vits2/configs/mb_istft_vits2_multi.json:29-72)."""
    hp = HParams()
    hp.abi_version = VITS_ABI_VERSION
    hp.n_vocab = n_vocab
    hp.hidden_channels = 192
    hp.inter_channels = 192
    hp.filter_channels = 768
    hp.n_heads = 2
    hp.n_layers = 6
    hp.kernel_size = 3
    hp.window_size = 4
    hp.gin_channels = 256
    hp.n_speakers = 200
    hp.enc_cond_layer = 2
    hp.dp_filter_channels = 256
    hp.dp_kernel_size = 3
    hp.dp_n_flows = 4
    hp.dp_num_bins = 10
    hp.dp_dds_layers = 3
    hp.flow_n_flows = 4
    hp.flow_wn_layers = 4
    hp.flow_kernel_size = 5
    hp.flow_dilation_rate = 1
    hp.dec_type = 0
    hp.dec_initial_channel = 512
    hp.n_ups = 2
    hp.up_rates[0], hp.up_rates[1] = 4, 4
    hp.up_kernels[0], hp.up_kernels[1] = 16, 16
    hp.n_resk = 3
    for i, k in enumerate((3, 7, 11)):
        hp.res_kernels[i] = k
    hp.n_resd = 3
    for i in range(3):
        for j, d in enumerate((1, 3, 5)):
            hp.res_dilations[i][j] = d
    hp.subbands = 4
    hp.istft_n_fft = 16
    hp.istft_hop = 4
    hp.pqmf_taps = 62
    hp.pqmf_cutoff = 0.15
    hp.pqmf_beta = 9.0
    hp.dp_tail_bound = 5.0
    hp.sampling_rate = 22050
    hp.hop_length = 256
    return hp


def tiny_hparams(n_vocab=20):
    """A scaled-down graph of the same family for fast CPU tests
    (channels stay multiples of 32 so every MFMA tile path is exercised)."""
    hp = default_hparams(n_vocab)
    hp.hidden_channels = 64
    hp.inter_channels = 64
    hp.filter_channels = 128
    hp.n_layers = 3
    hp.gin_channels = 32
    hp.n_speakers = 5
    # dp filter (256), flow depth (4x4) and flow kernel (5) are hard-coded in the
    # reference's SynthesizerTrn (models.py:1609-1625), so the tiny graph keeps them
    # and stays constructible by the reference for golden generation.
    hp.dec_initial_channel = 128
    return hp


def plain_hparams(n_vocab=20):
    """Tiny graph with the plain HiFi-GAN `Generator` decoder (models.py:845-898) and the classic V1
    upsampling [8,8,2,2] / kernels [16,16,4,4] — SURVEY.md §8a row a21."""
    hp = tiny_hparams(n_vocab)
    hp.dec_type = 1
    hp.dec_initial_channel = 512
    hp.n_ups = 4
    for i, (u, k) in enumerate(((8, 16), (8, 16), (2, 4), (2, 4))):
        hp.up_rates[i], hp.up_kernels[i] = u, k
    return hp


def hifigan_v1_vocoder_hparams():
    """Vocoder-only blob (n_vocab = 0): the HiFi-GAN V1 generator bundled with StableTTS
    (training/stabletts/matcha/hifigan/models.py:148-199, config.py v1) that the multistream export wraps as
    `vocoder.decode(mel)` (matcha/onnx/export.py:28-32): 80 mel channels in, ups [8,8,2,2], conv_post WITH bias,
    no speaker conditioning.  Only the decoder stage is available on such a model."""
    hp = default_hparams(0)
    hp.n_vocab = 0
    hp.n_speakers = 0
    hp.gin_channels = 0
    hp.inter_channels = 80
    hp.dec_type = 1
    hp.dec_initial_channel = 512
    hp.n_ups = 4
    for i, (u, k) in enumerate(((8, 16), (8, 16), (2, 4), (2, 4))):
        hp.up_rates[i], hp.up_kernels[i] = u, k
    return hp


# --------------------------------------------------------------------------- #
# tensor inventory
# --------------------------------------------------------------------------- #

def tensor_specs(hp):
    """Ordered [(name, shape, kind, fan_in)] for every tensor on the inference
    path.  kind selects the synthetic init."""
    H, I, F = hp.hidden_channels, hp.inter_channels, hp.filter_channels
    G = hp.gin_channels
    nh = hp.n_heads
    dk = H // nh
    W = 2 * hp.window_size + 1
    specs = []

    def conv(name, co, ci, k, bias=True, gain=1.0):
        specs.append((name + ".weight", (co, ci, k), "w", ci * k, gain))
        if bias:
            specs.append((name + ".bias", (co,), "b", 0, 1.0))

    def ln(name, c):
        specs.append((name + ".gamma", (c,), "gamma", 0, 1.0))
        specs.append((name + ".beta", (c,), "beta", 0, 1.0))

    def encoder(prefix, n_layers, filt, k):
        for i in range(n_layers):
            a = f"{prefix}.attn_layers.{i}"
            specs.append((a + ".emb_rel_k", (1, W, dk), "rel", dk, 1.0))
            specs.append((a + ".emb_rel_v", (1, W, dk), "rel", dk, 1.0))
            for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
                conv(f"{a}.{n}", H, H, 1)
        for i in range(n_layers):
            ln(f"{prefix}.norm_layers_1.{i}", H)
        for i in range(n_layers):
            conv(f"{prefix}.ffn_layers.{i}.conv_1", filt, H, k)
            conv(f"{prefix}.ffn_layers.{i}.conv_2", H, filt, k)
        for i in range(n_layers):
            ln(f"{prefix}.norm_layers_2.{i}", H)

    def ddsconv(prefix, c, k, n):
        for i in range(n):
            specs.append((f"{prefix}.convs_sep.{i}.weight", (c, 1, k), "w", k, 1.0))
            specs.append((f"{prefix}.convs_sep.{i}.bias", (c,), "b", 0, 1.0))
        for i in range(n):
            conv(f"{prefix}.convs_1x1.{i}", c, c, 1)
        for i in range(n):
            ln(f"{prefix}.norms_1.{i}", c)
        for i in range(n):
            ln(f"{prefix}.norms_2.{i}", c)

    acoustic = hp.n_vocab > 0  # n_vocab == 0: vocoder-only blob, decoder tensors only
    if acoustic:
        # text encoder (models.py:283-326)
        specs.append(("enc_p.emb.weight", (hp.n_vocab, H), "emb", H, 1.0))
        encoder("enc_p.encoder", hp.n_layers, F, hp.kernel_size)
        if hp.enc_cond_layer >= 0 and G > 0:
            specs.append(("enc_p.encoder.spk_emb_linear.weight", (H, G), "w", G, 1.0))
            specs.append(("enc_p.encoder.spk_emb_linear.bias", (H,), "b", 0, 1.0))
        conv("enc_p.proj", 2 * I, H, 1, gain=0.5)
        if hp.bert_dim > 0:  # BERT-conditioned flavour (vosk_tts/synth.py:88-99): 1x1 projection of the "bert" feed
            conv("enc_p.bert_proj", H, hp.bert_dim, 1, gain=0.5)

    # decoder (models.py:974-1014 / 845-871)
    C0 = hp.dec_initial_channel
    conv("dec.conv_pre", C0, I, 7)
    ch = C0
    for i in range(hp.n_ups):
        # ConvTranspose1d weight is [C_in, C_out, K] (models.py:986-990)
        specs.append((f"dec.ups.{i}.weight", (ch, ch // 2, hp.up_kernels[i]), "w", ch * hp.up_kernels[i] // hp.up_rates[i], 1.0))
        specs.append((f"dec.ups.{i}.bias", (ch // 2,), "b", 0, 1.0))
        ch //= 2
        for j in range(hp.n_resk):
            k = hp.res_kernels[j]
            rb = f"dec.resblocks.{i * hp.n_resk + j}"
            for d in range(hp.n_resd):
                conv(f"{rb}.convs1.{d}", ch, ch, k, gain=0.7)
            for d in range(hp.n_resd):
                conv(f"{rb}.convs2.{d}", ch, ch, k, gain=0.7)
    if hp.dec_type == 0:
        conv("dec.subband_conv_post", hp.subbands * (hp.istft_n_fft + 2), ch, 7, bias=False, gain=0.5)
    else:
        # VITS' Generator: no conv_post bias (models.py:866); StableTTS' HiFi-GAN: bias (hifigan/models.py:176)
        conv("dec.conv_post", 1, ch, 7, bias=not acoustic, gain=0.5)
        if G > 0 and hp.n_speakers > 1:
            conv("dec.cond", C0, G, 1)  # Generator.cond (models.py:869-870)
    if not acoustic:
        return specs

    # flow (models.py:329-396, 630-762); only even indices carry weights
    for f in range(hp.flow_n_flows):
        p = f"flow.flows.{2 * f}"
        conv(p + ".pre", H, I // 2, 1)
        encoder(p + ".pre_transformer", 1, H, hp.flow_kernel_size)
        for i in range(hp.flow_wn_layers):
            conv(f"{p}.enc.in_layers.{i}", 2 * H, H, hp.flow_kernel_size)
        for i in range(hp.flow_wn_layers):
            rs = 2 * H if i < hp.flow_wn_layers - 1 else H
            conv(f"{p}.enc.res_skip_layers.{i}", rs, H, 1, gain=0.7)
        if G > 0:
            conv(p + ".enc.cond_layer", 2 * H * hp.flow_wn_layers, G, 1)
        conv(p + ".post", I // 2, H, 1, gain=0.5)

    # stochastic duration predictor, reverse path only (models.py:23-63,93-101):
    # flows[0] ElementwiseAffine, flows[2k+1] ConvFlow for k>=1 (flows[1] is skipped, :94-95)
    D = hp.dp_filter_channels
    specs.append(("dp.flows.0.m", (2, 1), "small", 0, 1.0))
    specs.append(("dp.flows.0.logs", (2, 1), "small", 0, 1.0))
    for k in range(1, hp.dp_n_flows):
        p = f"dp.flows.{2 * k + 1}"
        conv(p + ".pre", D, 1, 1)
        ddsconv(p + ".convs", D, hp.dp_kernel_size, hp.dp_dds_layers)
        conv(p + ".proj", 3 * hp.dp_num_bins - 1, D, 1, gain=4.0)
    conv("dp.pre", D, H, 1)
    conv("dp.proj", D, D, 1)
    ddsconv("dp.convs", D, hp.dp_kernel_size, hp.dp_dds_layers)
    if G > 0:
        conv("dp.cond", D, G, 1)
    if hp.n_speakers > 1:
        specs.append(("emb_g.weight", (hp.n_speakers, G), "emb1", G, 1.0))
    return specs


def _rng(name, seed):
    key = np.array([zlib.crc32(name.encode()), seed & 0xFFFFFFFF], dtype=np.uint64)
    return np.random.Generator(np.random.Philox(key=key))


def make_synthetic_weights(hp, seed=1234, heavy_sigma=0.0):
    """name -> float32 ndarray.  Fan-in scaled uniform so activations stay O(1).

    heavy_sigma > 0: the "heavy-tailed" variant for dynamic-range tests -- every weight matrix gets per-row (first-axis) scales that
    are log-normal with that sigma, normalised to unit mean square (so the average gain of a layer is unchanged while individual
    channels differ by an order of magnitude, as weight-normed trained layers do; tests/golden/full_heavy.npz)."""
    return synthetic_from_specs(tensor_specs(hp), seed, heavy_sigma)


def synthetic_from_specs(specs, seed=1234, heavy_sigma=0.0):
    out = {}
    for name, shape, kind, fan_in, gain in specs:
        r = _rng(name, seed)
        u = r.random(size=shape, dtype=np.float64) * 2.0 - 1.0  # U(-1,1)
        if kind == "w":
            a = gain * np.sqrt(3.0 / max(fan_in, 1))
            t = u * a
            if heavy_sigma > 0 and len(shape) >= 2 and shape[0] > 1:
                g = np.exp(heavy_sigma * _rng(name + ":heavy", seed).standard_normal(shape[0]))
                g /= np.sqrt(np.mean(g * g))
                t = t * g.reshape((-1,) + (1,) * (len(shape) - 1))
        elif kind == "b":
            t = u * 0.05
        elif kind == "gamma":
            t = 1.0 + 0.1 * u
        elif kind == "beta":
            t = 0.05 * u
        elif kind == "rel":
            t = u * np.sqrt(3.0) * fan_in ** -0.5
        elif kind == "emb":
            t = u * np.sqrt(3.0) * fan_in ** -0.5  # std = H^-0.5 as nn.init.normal_ at models.py:304
        elif kind == "emb1":
            t = u * np.sqrt(3.0)
        elif kind == "small":
            t = u * 0.2
            if name == "dp.flows.0.m":
                t = t - 1.0  # logw = (z - m) * exp(-logs): centres free-running durations near e ~ 3 frames/token
        else:
            raise ValueError(kind)
        out[name] = np.ascontiguousarray(t.astype(np.float32))
    return out


# --------------------------------------------------------------------------- #
# blob I/O
# --------------------------------------------------------------------------- #

def validate_hparams(hp):
    """The checks vits_create makes before it divides by a rate or sizes a buffer from hop_length (engine.hip load_decoder):
    the decoder writes T_y * prod(up_rates) [* istft_hop * subbands] samples per item into buffers of T_y * hop_length."""
    if not isinstance(hp, HParams):
        return
    rate = 1
    for i in range(hp.n_ups):
        u, k = hp.up_rates[i], hp.up_kernels[i]
        if u <= 0 or k < u:
            raise ValueError(f"decoder stage {i}: upsample rate {u} / kernel {k} invalid")
        rate *= u
    if hp.dec_type == 0:
        if hp.subbands <= 0 or hp.istft_hop <= 0 or hp.istft_n_fft <= 0 or hp.istft_n_fft % hp.istft_hop:
            raise ValueError("iSTFT / PQMF parameters invalid")
        rate *= hp.istft_hop * hp.subbands
    if hp.conv_precision not in (0, 1):
        raise ValueError(f"conv_precision {hp.conv_precision}: 0 = fp32, 1 = split-bf16 decoder ResBlock convs")
    if rate != hp.hop_length:
        raise ValueError(f"decoder produces {rate} samples per frame but hop_length is {hp.hop_length}: upsample_rates / "
                         "gen_istft_hop_size / subbands / hop_length are inconsistent")


def pack_blob(hp, tensors, magic=MAGIC):
    validate_hparams(hp)
    names = list(tensors.keys())
    n = len(names)
    head = magic + struct.pack("<I", ctypes.sizeof(type(hp))) + bytes(hp) + struct.pack("<I", n)
    table_bytes = n * ctypes.sizeof(BlobEntry)
    off = len(head) + table_bytes
    off = (off + 63) // 64 * 64
    entries = []
    chunks = []
    cur = off
    for nm in names:
        a = np.ascontiguousarray(tensors[nm], dtype=np.float32)
        e = BlobEntry()
        e.name = nm.encode()
        e.ndim = a.ndim
        for i, d in enumerate(a.shape):
            e.dims[i] = d
        e.offset = cur
        e.nelem = a.size
        entries.append(bytes(e))
        b = a.tobytes()
        pad = (-len(b)) % 64
        chunks.append(b + b"\0" * pad)
        cur += len(b) + pad
    table = b"".join(entries)
    pre = head + table
    pre += b"\0" * (off - len(pre))
    return pre + b"".join(chunks)


def unpack_blob(blob):
    return unpack_blob_generic(blob, HParams, MAGIC)


def unpack_blob_generic(blob, hp_type, magic):
    """(hparams struct, {name: ndarray}) of any blob of this container family (VITSW001 / STTSW001 / BERTW001)"""
    if blob[:8] != magic:
        raise ValueError(f"not a {magic.decode()} blob")
    (hb,) = struct.unpack_from("<I", blob, 8)
    if hb != ctypes.sizeof(hp_type):
        raise ValueError("hparams size mismatch")
    hp = hp_type.from_buffer_copy(blob[12:12 + hb])
    (n,) = struct.unpack_from("<I", blob, 12 + hb)
    pos = 16 + hb
    tensors = {}
    es = ctypes.sizeof(BlobEntry)
    for i in range(n):
        e = BlobEntry.from_buffer_copy(blob[pos + i * es: pos + (i + 1) * es])
        shape = tuple(e.dims[j] for j in range(e.ndim))
        a = np.frombuffer(blob, dtype=np.float32, count=e.nelem, offset=e.offset).reshape(shape)
        tensors[e.name.decode()] = a
    return hp, tensors


def synthetic_blob(hp=None, seed=1234, heavy_sigma=0.0):
    hp = hp or default_hparams()
    return pack_blob(hp, make_synthetic_weights(hp, seed, heavy_sigma))


def save_blob(path, hp, tensors):
    with open(path, "wb") as f:
        f.write(pack_blob(hp, tensors))


def load_blob(path):
    with open(path, "rb") as f:
        return f.read()
