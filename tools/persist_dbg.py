"""persist on/off A/B of the full path (eager with injected noise, fast path with a seed)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib
lib = VitsLib()
m = lib.create(W.synthetic_blob(W.default_hparams(), 1234), 0)
rng = np.random.default_rng(0)
for Tx, d in ((50, 3), (16, 1), (37, 2)):
    ids = rng.integers(1, 62, size=(1, Tx)).astype(np.int64); lens = np.array([Tx], np.int64)
    dur = np.full((1, Tx), d, np.int32); Ty = Tx * d
    noise = rng.standard_normal((1, 192, Ty)).astype(np.float32)
    sc = np.array([0.8, 1.0, 0.8], np.float32)
    out = {}
    for mask in (0, 1, 2, 4, 7):
        lib.lib.vits_debug_persist(mask)
        a, l = m.synthesize(ids, lens, sc, [2], noise_prior=noise, forced_durations=dur)
        b, l2 = m.synthesize(ids, lens, sc, [2], forced_durations=dur, seed=5)
        out[mask] = (a, b)
    for mask in (1, 2, 4, 7):
        ea = np.abs(out[mask][0] - out[0][0]).max() / np.abs(out[0][0]).max()
        eb = np.abs(out[mask][1] - out[0][1]).max() / np.abs(out[0][1]).max()
        print(f"Tx={Tx} Ty={Ty} mask {mask}: eager rel err {ea:.2e}   fast path rel err {eb:.2e}")
