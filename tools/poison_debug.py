import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib
lib = VitsLib()
lib.lib.vits_debug_poison_workspace(1)
model = lib.create(W.synthetic_blob(W.default_hparams(), 1234), 0)
rng = np.random.default_rng(123)
B = 5
lengths = rng.integers(3, 41, size=B).astype(np.int64); Tx = int(lengths.max())
ids = rng.integers(1, 62, size=(B, Tx)).astype(np.int64)
sid = rng.integers(0, 200, size=B).astype(np.int64)
dur = rng.integers(0, 5, size=(B, Tx)).astype(np.int32)
a, ol = model.synthesize(ids, lengths, np.array([0.667, 1.0, 0.8], np.float32), sid, forced_durations=dur, seed=5)
print("lengths", lengths, "olen frames", ol // 256, "T_y", a.shape[1] // 256)
for b in range(B):
    bad = np.where(~np.isfinite(a[b]))[0]
    if len(bad):
        print(f"item {b}: {len(bad)} non-finite samples, frames {bad.min() / 256:.1f} .. {bad.max() / 256:.1f} (valid up to {ol[b] // 256})")
    else:
        print(f"item {b}: finite")
