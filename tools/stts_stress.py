"""Stress: sequential vs concurrent stts_synthesize on dirtied sessions; reports any mismatch."""
import os, sys, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from vosk_tts_amd import weights as W, weights_stts as S
from vosk_tts_amd.capi import VitsLib
from vosk_tts_amd.capi_stts import SttsModel

lib = VitsLib()
voc = lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234), 0)
m = SttsModel(lib, S.synthetic_blob(S.default_hparams(40, 7), 1234), voc)
rng = np.random.default_rng(61)
sc = np.array([0.8, 1.0, 0.8], np.float32)
# dirty the pools with a batch call and differently sized singles
ids = rng.integers(1, 40, size=(4, 5, 22)).astype(np.int64)
m.synthesize_batch(ids, np.array([22, 9, 15, 4]), sc, np.array([0, 3, 6, 1]), None, np.full((4, 22), 3.0, np.float32), seed=1, n_timesteps=2)
jobs = []
for k in range(6):
    T = int(rng.integers(6, 30))
    jobs.append((rng.integers(1, 40, size=(5, T)).astype(np.int64), np.full(T, 3.0, np.float32), int(rng.integers(0, 7)), 100 + k))
want = [m.synthesize(i, sc, s, None, p, seed=sd, n_timesteps=2, want_mel=True) for (i, p, s, sd) in jobs]
bad = 0
for rep in range(40):
    got = [None] * len(jobs)
    def work(k):
        i, p, s, sd = jobs[k]
        for _ in range(2):
            got[k] = m.synthesize(i, sc, s, None, p, seed=sd, n_timesteps=2, want_mel=True)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(len(jobs))]
    [t.start() for t in ths]; [t.join() for t in ths]
    for k in range(len(jobs)):
        for name, a, b in (("audio", want[k][0], got[k][0]), ("mel", want[k][1], got[k][1])):
            if not np.array_equal(a, b):
                bad += 1
                d = np.abs(a - b)
                idx = np.argwhere(d > 0)
                print(f"rep {rep} job {k} {name}: {len(idx)} of {a.size} differ, max {d.max():.3e}, first {idx[0]}, last {idx[-1]}, shape {a.shape}, nan {np.isnan(b).sum()}")
print("mismatches:", bad)
