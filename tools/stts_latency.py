"""Latency + per-op profile of the StableTTS (multistream) host entry point."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from vosk_tts_amd import weights as W, weights_stts as S
from vosk_tts_amd.capi import VitsLib
from vosk_tts_amd.capi_stts import SttsModel

lib = VitsLib()
voc = lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234), 0)
hp = S.default_hparams(62, 5)
m = SttsModel(lib, S.synthetic_blob(hp, 1234), voc)
rng = np.random.default_rng(0)
for Tx in (20, 50, 120):
    ids = rng.integers(1, 62, size=(5, Tx)).astype(np.int64)
    pde = np.full(Tx, 3.0, np.float32)  # pin 3 frames per symbol like the VITS bench
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    for want_audio in (True, False):
        for _ in range(3):
            a, mel = m.synthesize(ids, sc, 2, None, pde, seed=1, want_audio=want_audio)
        t = []
        for i in range(10):
            t0 = time.perf_counter(); a, mel = m.synthesize(ids, sc, 2, None, pde, seed=i, want_audio=want_audio); t.append(time.perf_counter() - t0)
        sec = mel.shape[1] * 256 / 22050
        print(f"T_x={Tx:4d} T_y={mel.shape[1]:5d} {'mel+vocoder' if want_audio else 'mel only   '}: median {np.median(t)*1e3:7.2f} ms -> {sec/np.median(t):7.1f}x real-time")
