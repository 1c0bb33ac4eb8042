#!/usr/bin/env python3
"""per-step host times of the m2 bench loop (stts_synthesize, 50 symbols): which calls stall and for how long.
    python tools/m2_steps.py [steps=120] [gc=1]"""
import gc, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from vosk_tts_amd import weights as W, weights_stts as S  # noqa: E402
from vosk_tts_amd.capi import VitsLib  # noqa: E402
from vosk_tts_amd.capi_stts import SttsModel  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
if len(sys.argv) > 2 and sys.argv[2] == "0":
    gc.disable()
lib = VitsLib()
voc = lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234), 0)
hp = S.default_hparams(62, 5)
model = SttsModel(lib, S.synthetic_blob(hp, 1234), voc, 0)
rng = np.random.default_rng(1234)
ids = rng.integers(1, 62, size=(5, 50)).astype(np.int64)
pde = np.full(50, 3.0, np.float32)
scales = np.array([0.8, 1.0, 0.8], np.float32)
t00 = time.perf_counter()
ts = []
for i in range(steps):
    t0 = time.perf_counter()
    a = model.synthesize(ids, scales, 2, None, pde, seed=7 + i, want_mel=False)[0]
    ts.append((time.perf_counter() - t0) * 1e3)
med = float(np.median(ts))
print(f"median {med:.3f} ms, mean {np.mean(ts):.3f}, total {sum(ts):.1f} ms over {steps} calls")
acc = 0.0
for i, t in enumerate(ts):
    if t > 1.5 * med:
        print(f"  call {i}: {t:.2f} ms (at {acc:.0f} ms into the loop)")
    acc += t
model.close()
