"""Word-embedding BERT encoder of the BERT-conditioned flavours (vosk_tts/model.py:59-63, synth.py:25-44): the
`bert/model.onnx` that training/stabletts/matcha/onnx/bert-export.py exports from a HuggingFace `BertModel` and whose
output is `hidden_states[-3]` (the activations after layer num_layers - 2).

Blob "BERTW001": `struct bert_hparams` (include/stts_mi355.h) + tensors named like `BertModel.state_dict()`
(the pooler and the two layers after the tapped one are not part of the path)."""
import ctypes

from . import weights as W

MAGIC = b"BERTW001"
BERT_ABI_VERSION = 1


class BertHParams(ctypes.Structure):
    _fields_ = [
        ("abi_version", ctypes.c_int32),
        ("vocab_size", ctypes.c_int32),
        ("hidden", ctypes.c_int32),          # 768
        ("n_layers", ctypes.c_int32),        # layers in the checkpoint (12)
        ("out_layers", ctypes.c_int32),      # layers actually run: hidden_states[-3] -> n_layers - 2
        ("n_heads", ctypes.c_int32),         # 12
        ("intermediate", ctypes.c_int32),    # 3072
        ("max_position", ctypes.c_int32),    # 512
        ("type_vocab", ctypes.c_int32),      # 2
        ("ln_eps", ctypes.c_float),          # 1e-12
    ]


def base_hparams(vocab_size=120):
    """rubert-base geometry (BERT-base): 12 x 768, 12 heads, 3072; the export taps hidden_states[-3]"""
    hp = BertHParams()
    hp.abi_version = BERT_ABI_VERSION
    hp.vocab_size, hp.hidden, hp.n_layers, hp.n_heads, hp.intermediate = vocab_size, 768, 12, 12, 3072
    hp.out_layers = hp.n_layers - 2
    hp.max_position, hp.type_vocab, hp.ln_eps = 512, 2, 1e-12
    return hp


def small_hparams(vocab_size=120, hidden=768, n_layers=4):
    """same width (the acoustic model's bert_proj takes 768), fewer layers: for tests and toy voices"""
    hp = base_hparams(vocab_size)
    hp.hidden, hp.n_layers, hp.out_layers = hidden, n_layers, n_layers - 2
    hp.n_heads, hp.intermediate, hp.max_position = hidden // 64, 4 * hidden, 128
    return hp


def tensor_specs(hp):
    H, F = hp.hidden, hp.intermediate
    specs = [("embeddings.word_embeddings.weight", (hp.vocab_size, H), "emb1", H, 1.0),
             ("embeddings.position_embeddings.weight", (hp.max_position, H), "emb1", H, 1.0),
             ("embeddings.token_type_embeddings.weight", (hp.type_vocab, H), "emb1", H, 1.0),
             ("embeddings.LayerNorm.weight", (H,), "gamma", 0, 1.0), ("embeddings.LayerNorm.bias", (H,), "beta", 0, 1.0)]

    def linear(name, co, ci):
        specs.append((name + ".weight", (co, ci), "w", ci, 1.0))
        specs.append((name + ".bias", (co,), "b", 0, 1.0))

    for i in range(hp.out_layers):
        p = f"encoder.layer.{i}"
        for n in ("query", "key", "value"):
            linear(f"{p}.attention.self.{n}", H, H)
        linear(f"{p}.attention.output.dense", H, H)
        specs.append((f"{p}.attention.output.LayerNorm.weight", (H,), "gamma", 0, 1.0))
        specs.append((f"{p}.attention.output.LayerNorm.bias", (H,), "beta", 0, 1.0))
        linear(f"{p}.intermediate.dense", F, H)
        linear(f"{p}.output.dense", H, F)
        specs.append((f"{p}.output.LayerNorm.weight", (H,), "gamma", 0, 1.0))
        specs.append((f"{p}.output.LayerNorm.bias", (H,), "beta", 0, 1.0))
    return specs


def make_synthetic_weights(hp, seed=1234):
    return W.synthetic_from_specs(tensor_specs(hp), seed)


def pack_blob(hp, tensors):
    return W.pack_blob(hp, tensors, magic=MAGIC)


def synthetic_blob(hp=None, seed=1234):
    hp = hp or base_hparams()
    return pack_blob(hp, make_synthetic_weights(hp, seed))
