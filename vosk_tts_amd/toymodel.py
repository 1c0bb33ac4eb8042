"""Writes a self-contained model directory (model.vitsw + dictionary + config.json) with seeded
synthetic weights, in the layout vosk_tts.Model reads (vosk_tts/model.py:46-63).  Used by tests,
smoke() and the docs example, because the real voices cannot be downloaded offline."""
import json
import os

from . import weights as W

_CONS = ["b", "v", "g", "d", "z", "k", "l", "m", "n", "p", "r", "s", "t", "f", "h"]
PHONEMES = (["_", "^", "$", " ", ",", ".", "?", "!", ";", ":", '"', "(", ")", "-"] + _CONS + [c + "j" for c in _CONS] +
            ["zh", "c", "ch", "sh", "sch", "j"] + [v + s for v in "aoueiy" for s in "01"])


def phoneme_id_map():
    return {p: i for i, p in enumerate(PHONEMES)}


def write_toy_model(path, hp=None, seed=1234, dictionary=None, inference=None, bert=False, no_blank=0):
    """bert=True: a BERT-conditioned VITS voice (vosk_tts/synth.py:88-99): hp.bert_dim = 768, enc_p.bert_proj in the blob,
    bert/vocab.txt + bert/model.bertw next to it; no_blank selects g2p_noblank (config "no_blank")."""
    os.makedirs(path, exist_ok=True)
    hp = hp or W.default_hparams(n_vocab=len(PHONEMES))
    if bert:
        hp.bert_dim = 768
        write_bert_dir(os.path.join(path, "bert"), seed)
    if hp.n_vocab < len(PHONEMES):
        raise ValueError("n_vocab too small for the phoneme inventory")
    W.save_blob(os.path.join(path, "model.vitsw"), hp, W.make_synthetic_weights(hp, seed))
    dictionary = dictionary or {"привет": [(1.0, "p rj i0 vj e1 t")], "мир": [(0.4, "mj i0 r"), (0.9, "mj i1 r")]}
    with open(os.path.join(path, "dictionary"), "w", encoding="utf-8") as f:
        for word, prons in dictionary.items():
            for prob, ph in prons:
                f.write(f"{word} {prob} {ph}\n")
    cfg = {"audio": {"sample_rate": hp.sampling_rate},
           "inference": inference or {"noise_level": 0.8, "speech_rate": 1.0, "duration_noise_level": 0.8, "scale": 1.0},
           "phoneme_id_map": phoneme_id_map(), "num_speakers": hp.n_speakers, "model_type": "vits", "no_blank": no_blank}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, ensure_ascii=False)
    return path


# ---- multistream (StableTTS / Matcha) toy voice ------------------------------------------------------------------
_MS_PUNCT = ["_", " ", "^", "$", ",", ".", "...", "?", "!", ";", ":", "(", ")", "-"]
_MS_BASE = [p for p in PHONEMES if p not in _MS_PUNCT and p != '"']


def multistream_phoneme_id_map():
    """ids 0/1 stay free for the in-quote stream, which the graph consumes as a raw id (synth.py:343)"""
    syms = _MS_PUNCT + [b + suf for b in _MS_BASE for suf in ("", "_B", "_I", "_E", "_S")]  # "" = multistream_v1 (no word positions)
    return {p: i + 2 for i, p in enumerate(syms)}


_RU = "абвгдеёжзийклмнопрстуфхцчшщъыьэюя"
BERT_VOCAB = (["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"] + list(",.?!;:\"()-") + list(_RU) + ["##" + c for c in _RU] +
              ["привет", "мир", "да", "нет"])


def write_bert_dir(path, seed=1234, n_layers=4):
    """bert/vocab.txt (WordPiece, model.py:59-60) + bert/model.bertw (768-wide synthetic encoder, weights_bert.py)"""
    from . import weights_bert as BW

    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "vocab.txt"), "w", encoding="utf-8") as f:
        f.write("\n".join(BERT_VOCAB) + "\n")
    with open(os.path.join(path, "model.bertw"), "wb") as f:
        f.write(BW.synthetic_blob(BW.small_hparams(len(BERT_VOCAB), 768, n_layers), seed))


def write_toy_multistream_model(path, seed=1234, n_spks=5, inference=None, model_type="multistream_v2", with_bert=False):
    """model.sttsw + vocoder.vitsw + dictionary + config.json; with_bert adds bert/ (tokenizer vocabulary + encoder)"""
    from . import weights_stts as S

    os.makedirs(path, exist_ok=True)
    idmap = multistream_phoneme_id_map()
    hp = S.default_hparams(n_vocab=max(idmap.values()) + 1, n_spks=n_spks)
    with open(os.path.join(path, "model.sttsw"), "wb") as f:
        f.write(S.synthetic_blob(hp, seed))
    W.save_blob(os.path.join(path, "vocoder.vitsw"), W.hifigan_v1_vocoder_hparams(), W.make_synthetic_weights(W.hifigan_v1_vocoder_hparams(), seed))
    with open(os.path.join(path, "dictionary"), "w", encoding="utf-8") as f:
        f.write("привет 1.0 p rj i0 vj e1 t\nмир 0.9 mj i1 r\n")
    cfg = {"audio": {"sample_rate": hp.sampling_rate},
           "inference": inference or {"noise_level": 0.8, "speech_rate": 1.0, "duration_noise_level": 0.8, "scale": 1.0},
           "phoneme_id_map": idmap, "num_speakers": n_spks, "model_type": model_type}
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(cfg, f, ensure_ascii=False)
    if with_bert:
        write_bert_dir(os.path.join(path, "bert"), seed)
    return path


def multistream_dir_to_reference_layout(path):
    """Rewrite a toy multistream directory into the REFERENCE's file layout (vosk_tts/model.py:46,62): model.onnx with the
    acoustic model under "matcha." and the HiFi-GAN generator under "vocoder." (MatchaWithVocoder, matcha/onnx/export.py:21-32),
    bert/model.onnx; the blobs are removed.  Files come from onnx_import.write_minimal_onnx (initializers only, no graph), so
    what the graph would tell (n_timesteps, mel statistics) stays at the defaults the toy blobs use."""
    from . import onnx_import as oi
    from . import weights_bert as BW
    from . import weights_stts as S

    _hp, t = W.unpack_blob_generic(open(os.path.join(path, "model.sttsw"), "rb").read(), S.SttsHParams, S.MAGIC)
    _vhp, vt = W.unpack_blob(open(os.path.join(path, "vocoder.vitsw"), "rb").read())
    tensors = {"matcha." + k: v for k, v in t.items()}
    tensors.update({"vocoder." + k[len("dec."):]: v for k, v in vt.items()})
    oi.write_minimal_onnx(os.path.join(path, "model.onnx"), tensors)
    os.remove(os.path.join(path, "model.sttsw"))
    os.remove(os.path.join(path, "vocoder.vitsw"))
    b = os.path.join(path, "bert", "model.bertw")
    if os.path.exists(b):
        _bhp, bt = W.unpack_blob_generic(open(b, "rb").read(), BW.BertHParams, BW.MAGIC)
        oi.write_minimal_onnx(os.path.join(path, "bert", "model.onnx"), bt)
        os.remove(b)
    return path
