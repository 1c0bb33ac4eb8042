"""Wall-clock latency of the host-buffer API (what Synth.synth_audio brackets, vosk_tts/synth.py:122-131)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib

lib = VitsLib()
model = lib.create(W.synthetic_blob(W.default_hparams(), 1234), 0)
rng = np.random.default_rng(1234)
for Tx in (10, 50, 200):
    ids = rng.integers(1, 62, size=(1, Tx)).astype(np.int64)
    lens = np.array([Tx], np.int64); sid = np.array([2], np.int64); scales = np.array([0.8, 1.0, 0.8], np.float32)
    for mode in ("free", "pinned3"):
        kw = {} if mode == "free" else {"forced_durations": np.full((1, Tx), 3, np.int32)}
        for _ in range(3):
            a, ol = model.synthesize(ids, lens, scales, sid, seed=1, **kw)
        t = []
        for i in range(20):
            t0 = time.perf_counter()
            a, ol = model.synthesize(ids, lens, scales, sid, seed=i, **kw)
            pcm = np.clip(a.squeeze() * 32767.0, -32767.0, 32767.0).astype("int16")
            t.append(time.perf_counter() - t0)
        sec = ol[0] / 22050.0
        print(f"T_x={Tx:4d} {mode:8s}: audio {sec:6.2f} s  median {np.median(t)*1e3:7.2f} ms  min {min(t)*1e3:7.2f} ms  -> {sec/np.median(t):8.1f}x real-time")
