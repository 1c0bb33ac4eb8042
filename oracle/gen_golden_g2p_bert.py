#!/usr/bin/env python3
"""Generate tests/golden/g2p_bert.npz: outputs of the REFERENCE's own BERT-conditioned VITS front-ends
vosk_tts.Synth.g2p (vosk_tts/synth.py:152-188) and Synth.g2p_noblank (:190-220) on fixed sentences, with the toy dictionary
and id map of vosk_tts_amd.toymodel as the model data and the integers 0, 1, 2, ... standing in for the per-word BERT rows
(so the fixture records WHICH word vector each phoneme position receives).  TEST INFRASTRUCTURE, container-only:
synth.py is loaded from /root/reference by file path with onnxruntime stubbed out.   python oracle/gen_golden_g2p_bert.py"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = os.environ.get("VOSK_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

SENTENCES = ['Прив+ет, "м+ир" - да... Нет!', "м+ир", "Да? Нет. (Мож+ет б+ыть): хорош+о; ладно - пок+а!", '"Прив+ет" сказ+ал +он, и уш+ёл...',
             "од+ин -  дв+а -тр+и", "чт+о... чт+о?! д+а.", " прив+ет  м+ир "]


def main():
    from vosk_tts_amd.toymodel import phoneme_id_map

    sys.modules.setdefault("onnxruntime", types.ModuleType("onnxruntime"))
    pkg = types.ModuleType("vosk_tts_ref")
    pkg.__path__ = [os.path.join(REF, "vosk_tts")]
    sys.modules["vosk_tts_ref"] = pkg
    for name in ("g2p", "synth"):
        spec = importlib.util.spec_from_file_location(f"vosk_tts_ref.{name}", os.path.join(REF, "vosk_tts", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"vosk_tts_ref.{name}"] = mod
        spec.loader.exec_module(mod)
    Synth = sys.modules["vosk_tts_ref.synth"].Synth
    model = types.SimpleNamespace(dic={"привет": "p rj i0 vj e1 t", "мир": "mj i1 r"}, config={"phoneme_id_map": phoneme_id_map()}, tokenizer=None)
    synth = Synth(model)
    emb = list(range(200))  # "row i of get_word_bert"; the reference indexes it by word and uses [-1] for '$'
    out = {"sentences": np.array(SENTENCES)}
    for fn in ("g2p", "g2p_noblank"):
        ids, rows, offs = [], [], [0]
        for s in SENTENCES:
            with contextlib.redirect_stdout(io.StringIO()):
                i, e = getattr(synth, fn)(s, emb)
            assert len(i) == len(e)
            ids.extend(i); rows.extend(e); offs.append(len(ids))
        out[fn + "_ids"] = np.array(ids, np.int64)
        out[fn + "_rows"] = np.array(rows, np.int64)
        out[fn + "_offsets"] = np.array(offs, np.int64)
    np.savez_compressed(os.path.join(OUT, "g2p_bert.npz"), **out)
    print("g2p_bert.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
