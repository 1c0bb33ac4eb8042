"""Hyper-parameters, tensor inventory, synthetic weights and blob packing for the StableTTS / Matcha acoustic model
(the `multistream_v*` flavours of vosk_tts/synth.py:64-87; reference: training/stabletts/matcha/models).

Blob "STTSW001": same container as the VITS blob (weights.pack_blob) with `struct stts_hparams`
(include/stts_mi355.h); tensor names are the reference `MatchaTTS.state_dict()` keys.  The mel `encoder.encoder.*`
stack is NOT part of the inference path (its outputs only feed `mel_enc`, matcha_tts.py:167-171,199) and is left out.
The vocoder (`vocoder.decode(mel)`, matcha/onnx/export.py:28-32) is a separate vocoder-only VITSW001 blob
(weights.hifigan_v1_vocoder_hparams).
"""
import ctypes

import numpy as np

from . import weights as W

MAGIC = b"STTSW001"
STTS_ABI_VERSION = 1


class SttsHParams(ctypes.Structure):
    """ctypes mirror of `struct stts_hparams` (include/stts_mi355.h)."""

    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("n_vocab", ctypes.c_int32),
        ("n_spks", ctypes.c_int32),
        ("spk_emb_dim", ctypes.c_int32),      # 128 (configs/model/matcha.yaml)
        ("n_feats", ctypes.c_int32),          # 80 mel channels
        ("emb_dim", ctypes.c_int32),          # 160: phoneme stream (text_encoder.py:100)
        ("punc_dim", ctypes.c_int32),         # 16: each of the 4 auxiliary streams (:104)
        ("bert_dim", ctypes.c_int32),         # 768
        ("bert_proj_dim", ctypes.c_int32),    # 32 (:108)
        ("enc_hidden", ctypes.c_int32),       # 256 = 160 + 4*16 + 32
        ("enc_filter", ctypes.c_int32),       # 1024
        ("enc_heads", ctypes.c_int32),        # 4
        ("enc_layers", ctypes.c_int32),       # 4
        ("enc_kernel", ctypes.c_int32),       # 3
        ("dp_out", ctypes.c_int32),           # 50 duration logits per token (:85)
        ("dec_hidden", ctypes.c_int32),       # 384 (flow_matching.py:300)
        ("dec_filter", ctypes.c_int32),       # 768
        ("dec_heads", ctypes.c_int32),        # 4
        ("dec_layers", ctypes.c_int32),       # 6 (long skip connections pair 0-5, 1-4, 2-3)
        ("dec_kernel", ctypes.c_int32),       # 3
        ("n_timesteps", ctypes.c_int32),      # Euler steps, exporter default 5 (onnx/export.py:111)
        ("guidance_scale", ctypes.c_float),   # 0.5 (flow_matching.py:61)
        ("mel_mean", ctypes.c_float),
        ("mel_std", ctypes.c_float),
        ("hop_length", ctypes.c_int32),       # 256
        ("sampling_rate", ctypes.c_int32),    # 22050
    ]


def default_hparams(n_vocab=62, n_spks=5):
    hp = SttsHParams()
    hp.abi_version = STTS_ABI_VERSION
    hp.n_vocab, hp.n_spks, hp.spk_emb_dim, hp.n_feats = n_vocab, n_spks, 128, 80
    hp.emb_dim, hp.punc_dim, hp.bert_dim, hp.bert_proj_dim = 160, 16, 768, 32
    hp.enc_hidden, hp.enc_filter, hp.enc_heads, hp.enc_layers, hp.enc_kernel, hp.dp_out = 256, 1024, 4, 4, 3, 50
    hp.dec_hidden, hp.dec_filter, hp.dec_heads, hp.dec_layers, hp.dec_kernel = 384, 768, 4, 6, 3
    hp.n_timesteps, hp.guidance_scale = 5, 0.5
    hp.mel_mean, hp.mel_std = -5.5, 2.1
    hp.hop_length, hp.sampling_rate = 256, 22050
    return hp


def tensor_specs(hp):
    """Ordered [(name, shape, kind, fan_in, gain)] of the inference path (reference state_dict names)."""
    specs = []
    G = hp.spk_emb_dim

    def conv(name, co, ci, k, gain=1.0):
        specs.append((name + ".weight", (co, ci, k), "w", ci * k, gain))
        specs.append((name + ".bias", (co,), "b", 0, 1.0))

    def linear(name, co, ci, gain=1.0):
        specs.append((name + ".weight", (co, ci), "w", ci, gain))
        specs.append((name + ".bias", (co,), "b", 0, 1.0))

    def dit_block(p, H, F, k):  # DiTConVBlock (diffusion_transformer.py:82-118)
        for n in ("conv_q", "conv_k", "conv_v", "conv_o"):
            conv(f"{p}.attn.{n}", H, H, 1)
        conv(f"{p}.mlp.conv_1", F, H, k)
        conv(f"{p}.mlp.conv_2", H, F, k)
        linear(f"{p}.adaLN_modulation.0", H, G)
        linear(f"{p}.adaLN_modulation.2", 6 * H, H, gain=0.5)  # the reference zero-inits this; trained models do not stay there

    specs.append(("spk_emb.weight", (hp.n_spks, G), "emb1", G, 1.0))
    specs.append(("dur_spk_emb.weight", (hp.n_spks, G), "emb1", G, 1.0))
    specs.append(("encoder.emb.weight", (hp.n_vocab, hp.emb_dim), "emb", hp.emb_dim, 1.0))
    specs.append(("encoder.punc_emb.weight", (hp.n_vocab, hp.punc_dim), "emb", hp.punc_dim, 1.0))
    linear("encoder.bert_proj.1", hp.bert_proj_dim, hp.bert_dim)
    H, F = hp.enc_hidden, hp.enc_filter
    for i in range(hp.enc_layers):
        dit_block(f"encoder.dp_encoder.encoder.{i}", H, F, hp.enc_kernel)
    conv("encoder.dp_encoder.proj", hp.dp_out, H, 1, gain=2.0)
    e = "decoder.estimator"
    Hd, Fd = hp.dec_hidden, hp.dec_filter
    linear(f"{e}.time_mlp.layer.0", Fd, Hd)
    linear(f"{e}.time_mlp.layer.2", Hd, Fd)
    conv(f"{e}.in_proj", Hd, Hd + hp.n_feats, 1)
    for i in range(hp.dec_layers):
        conv(f"{e}.blocks.{i}.time_fusion.film", 2 * Hd, Hd, 1, gain=0.5)
        dit_block(f"{e}.blocks.{i}.block", Hd, Fd, hp.dec_kernel)
    conv(f"{e}.final_proj", hp.n_feats, Hd, 1)
    conv(f"{e}.cond_proj.0", Fd, hp.enc_hidden, hp.dec_kernel)
    conv(f"{e}.cond_proj.2", Fd, Fd, hp.dec_kernel)
    conv(f"{e}.cond_proj.4", Hd, Fd, hp.dec_kernel)
    for j in range(hp.dec_layers // 2):
        conv(f"{e}.lsc_layers.{j}", Hd, 2 * Hd, hp.dec_kernel)
    specs.append(("fake_speaker", (1, G), "small", 0, 1.0))
    specs.append(("fake_content", (1, hp.enc_hidden, 1), "small", 0, 1.0))
    return specs


def make_synthetic_weights(hp, seed=1234):
    t = W.synthetic_from_specs(tensor_specs(hp), seed)
    # FiLM gamma should sit near 1 (gamma * x + beta, decoder.py:31-33): bias the first half of each film bias
    for i in range(hp.dec_layers):
        b = t[f"decoder.estimator.blocks.{i}.time_fusion.film.bias"]
        b[: hp.dec_hidden] += 1.0
    # durations are sum_k sigmoid(logit_k) over 50 logits (matcha_tts.py:147): centre the logits so a token lasts ~5 frames
    t["encoder.dp_encoder.proj.bias"] -= 2.2
    return t


def pack_blob(hp, tensors):
    return W.pack_blob(hp, tensors, magic=MAGIC)


def synthetic_blob(hp=None, seed=1234):
    hp = hp or default_hparams()
    return pack_blob(hp, make_synthetic_weights(hp, seed))
