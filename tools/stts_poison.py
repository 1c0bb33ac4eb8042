"""Uninitialised-read hunt: every STTS / BERT entry point with NaN-poisoned workspaces must give the unpoisoned result."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from vosk_tts_amd import weights as W, weights_stts as S, weights_bert as WB
from vosk_tts_amd.capi import VitsLib
from vosk_tts_amd.capi_stts import SttsModel, BertEncoder

lib = VitsLib()
voc = lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234), 0)
m = SttsModel(lib, S.synthetic_blob(S.default_hparams(40, 7), 1234), voc)
rng = np.random.default_rng(5)
sc = np.array([0.8, 1.0, 0.8], np.float32)

def run_all():
    out = {}
    for T in (1, 3, 7, 22, 45):
        ids = rng0.integers(1, 40, size=(5, T)).astype(np.int64)
        a, mel = m.synthesize(ids, sc, 3, None, np.full(T, 3.0, np.float32), seed=7, n_timesteps=2, want_mel=True)
        out[f"single{T}.audio"], out[f"single{T}.mel"] = a, mel
        a, mel = m.synthesize(ids, sc, 2, rng0.standard_normal((768, T)).astype(np.float32), None, seed=7, n_timesteps=2, want_mel=True)
        out[f"singleb{T}.audio"], out[f"singleb{T}.mel"] = a, mel
    ids = rng0.integers(1, 40, size=(4, 5, 22)).astype(np.int64)
    r = m.synthesize_batch(ids, np.array([22, 9, 15, 4]), sc, np.array([0, 3, 6, 1]), None, np.full((4, 22), 3.0, np.float32), seed=1, n_timesteps=2)
    out["batch.audio"], out["batch.len"] = r[0], r[1]
    r = m.synthesize_batch(ids[:, :, :13], np.array([1, 13, 2, 7]), sc, np.array([0, 3, 6, 1]), None, None, seed=1, n_timesteps=2)
    out["batch2.audio"], out["batch2.len"] = r[0], r[1]
    return out

rng0 = np.random.default_rng(5); base = run_all()
lib.lib.vits_debug_poison_workspace(1)
m2 = SttsModel(lib, S.synthetic_blob(S.default_hparams(40, 7), 1234), lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234), 0))
m_keep, m = m, m2   # fresh pools -> fresh poisoned workspaces
rng0 = np.random.default_rng(5); pois = run_all()
bad = 0
for k in base:
    a, b = np.asarray(base[k]), np.asarray(pois[k])
    if a.shape != b.shape or not np.array_equal(a, b):
        bad += 1
        d = np.abs(a.astype(np.float64) - b) if a.shape == b.shape else None
        print("MISMATCH", k, a.shape, b.shape, "nan:", np.isnan(b).sum() if b.dtype.kind == 'f' else '-', "max", None if d is None else np.nanmax(d))
print("poison mismatches:", bad, "of", len(base))
