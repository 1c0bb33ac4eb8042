#!/usr/bin/env python3
"""stts_bert_encode at sentence sizes (BERT-base geometry, synthetic weights): ms per call; run under rocprofv3 --kernel-trace --stats for the split.
    python tools/bert_profile.py [tokens=12] [reps=200]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vosk_tts_amd import weights_bert as BW  # noqa: E402
from vosk_tts_amd.capi import VitsLib  # noqa: E402
from vosk_tts_amd.capi_stts import BertEncoder  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
enc = BertEncoder(VitsLib(), BW.synthetic_blob(BW.small_hparams(300, 768, 12), 7))
ids = np.random.default_rng(3).integers(0, 300, size=T)
for _ in range(10): enc.encode(ids)
t0 = time.perf_counter()
for _ in range(reps): enc.encode(ids)
print(f"T={T}: {(time.perf_counter()-t0)/reps*1e3:.3f} ms per encode (10 of 12 layers run)")
enc.close()
