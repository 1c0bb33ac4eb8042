"""`Model` — mirror of vosk_tts.Model (vosk_tts/model.py:33-63) with the ONNX Runtime session
replaced by the MI355X-native engine.

A model directory holds (reference layout, model.py:46-63):
    model.vitsw   weight blob for the HIP engine (see vosk_tts_amd/weights.py) — or the reference's model.onnx,
                  whose initializers are imported on load (vosk_tts_amd/onnx_import.py); multistream voices:
                  model.sttsw + vocoder.vitsw, or their model.onnx with the vocoder embedded; bert/model.bertw or
                  the reference's bert/model.onnx
    dictionary    word prob phonemes... ; the highest-probability pronunciation wins (model.py:48-55)
    config.json   inference defaults, phoneme_id_map, model_type, no_blank (synth.py:50-56,64,88,177)
Attributes kept: .onnx (object with .run(None, feed)), .dic, .config, .tokenizer.
There is no network in this build: models are looked up locally only (same search path).
"""
import json
import logging
import os
import re
import sys
from pathlib import Path

MODEL_DIRS = [os.getenv("VOSK_MODEL_PATH"), Path("/usr/share/vosk"), Path.home() / "AppData/Local/vosk",
              Path.home() / ".cache/vosk"]


def _local_models():
    for directory in MODEL_DIRS:
        if directory is None or not Path(directory).exists():
            continue
        for name in sorted(os.listdir(directory)):
            yield Path(directory, name)


def list_models():
    for p in _local_models():
        print(p.name)


def list_languages():
    langs = set()
    for p in _local_models():
        m = re.match(r"vosk-model(-small)?-tts-([a-z]{2}(-[a-z]{2})?)", p.name) or re.match(r"vosk-model(-small)?-([a-z]{2}(-[a-z]{2})?)", p.name)
        if m:
            langs.add(m.group(2))
    for lang in sorted(langs):
        print(lang)


class Model:
    def __init__(self, model_path=None, model_name=None, lang=None, device=None):
        from .session import VitsSession

        if model_path is None:
            model_path = self.get_model_path(model_name, lang)
        else:
            model_path = Path(model_path)
        if device is None:
            device = int(os.getenv("VOSK_TTS_DEVICE", os.getenv("LOCAL_RANK", "0")))
        logging.info(f"Loading model from {model_path}")
        blob_path = model_path / "model.vitsw"
        if (model_path / "model.sttsw").exists():
            # a `multistream_v*` (StableTTS / Matcha) voice: acoustic blob + vocoder-only blob (vosk_tts_amd/weights_stts.py)
            from .session_stts import SttsSession

            with open(model_path / "model.sttsw", "rb") as f:
                blob = f.read()
            with open(model_path / "vocoder.vitsw", "rb") as f:
                vblob = f.read()
            self.onnx = SttsSession(blob, vblob, device=device)
        elif blob_path.exists():
            with open(blob_path, "rb") as f:
                blob = f.read()
        elif (model_path / "model.onnx").exists():
            # a reference model directory (model.py:46): pull the initializers out of the exported graph
            from . import weights as W
            from . import onnx_import as oi

            cfg, raw = {}, {}
            if (model_path / "config.json").exists():
                with open(model_path / "config.json") as f:
                    raw = json.load(f)
                cfg = dict(raw.get("model", {}), **raw.get("data", {})) if isinstance(raw, dict) else {}
            if str(raw.get("model_type", "")).startswith("multistream"):
                # StableTTS / Matcha export with the vocoder embedded (matcha/onnx/export.py:21-32)
                from . import weights_stts as S
                from .session_stts import SttsSession

                hp, tensors, voc = oi.import_stts_onnx(str(model_path / "model.onnx"), cfg)
                if voc is None:
                    raise NotImplementedError("model.onnx has no embedded vocoder: a mel-only export cannot produce audio")
                self.onnx = SttsSession(S.pack_blob(hp, tensors), W.pack_blob(*voc), device=device)
            else:
                hp, tensors = oi.import_onnx(str(model_path / "model.onnx"), cfg)
                blob = W.pack_blob(hp, tensors)
        else:
            raise FileNotFoundError(f"neither {blob_path} nor model.onnx found in {model_path}")
        if not hasattr(self, "onnx"):
            self.onnx = VitsSession(blob, device=device)

        self.dic = {}
        probs = {}
        with open(model_path / "dictionary", encoding="utf-8") as f:
            for line in f:
                items = line.split(maxsplit=2)
                if len(items) < 3:
                    continue
                prob = float(items[1])
                if probs.get(items[0], 0) < prob:
                    self.dic[items[0]] = items[2]
                    probs[items[0]] = prob

        with open(model_path / "config.json") as f:
            self.config = json.load(f)
        # BERT word embeddings (model.py:59-63): WordPiece tokenizer + encoder; the encoder weights are a BERTW001 blob
        # (vosk_tts_amd/weights_bert.py) instead of bert/model.onnx
        self.tokenizer = None
        if os.path.exists(model_path / "bert/vocab.txt"):
            bert_blob = None
            if os.path.exists(model_path / "bert/model.bertw"):
                with open(model_path / "bert/model.bertw", "rb") as f:
                    bert_blob = f.read()
            elif os.path.exists(model_path / "bert/model.onnx"):  # the reference's file (model.py:62): import its initializers
                from . import weights_bert as BW
                from .onnx_import import import_bert_onnx

                bert_blob = BW.pack_blob(*import_bert_onnx(str(model_path / "bert/model.onnx")))
            if bert_blob is not None:
                from tokenizers import BertWordPieceTokenizer

                from .capi_stts import BertEncoder

                self.tokenizer = BertWordPieceTokenizer(vocab=str(model_path / "bert/vocab.txt"), unk_token="[UNK]", lowercase=True)
                self.bert_onnx = BertEncoder(self.onnx._lib, bert_blob, device)
            else:
                logging.warning("bert/vocab.txt found without bert/model.bertw or bert/model.onnx: BERT conditioning disabled")

    def warmup(self, **kw):
        """Pre-builds workspaces, persistent programs and captured graphs of the common request sizes (VitsSession.warmup); a no-op for
        session types without it.  Not part of the reference API: call it once after loading when first-request latency matters."""
        fn = getattr(self.onnx, "warmup", None)
        return fn(**kw) if fn is not None else (0, 0.0)

    def get_model_path(self, model_name, lang):
        if model_name is None:
            return self.get_model_by_lang(lang)
        return self.get_model_by_name(model_name)

    def get_model_by_name(self, model_name):
        for p in _local_models():
            if p.name == model_name:
                return p
        print("model name %s does not exist" % (model_name))
        sys.exit(1)

    def get_model_by_lang(self, lang):
        for p in _local_models():
            if re.match(r"vosk-model(-small)?-{}".format(lang), p.name):
                return p
        print("lang %s does not exist" % (lang))
        sys.exit(1)
