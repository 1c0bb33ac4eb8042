import sys, os, numpy as np
sys.path.insert(0, "/root/repo")
import torch
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib
import importlib.util
spec = importlib.util.spec_from_file_location("bench", "/root/repo/bench.py"); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
lib = VitsLib(); hp = W.default_hparams(); m = lib.create(W.synthetic_blob(hp, 1234), 0)
print(lib.decoder_needs(hp))
ids, lengths, dur = bench.make_workload("c3", np.random.default_rng(1234))
B = ids.shape[0]; ylens = dur.sum(1)
a, l = m.synthesize(ids, lengths, np.array([0.8,1,0.8],np.float32), np.full(B,2,np.int64), forced_durations=dur, seed=7)
offs = []
for b in range(B):
    nz = np.nonzero(a[b])[0]
    offs.append(int(nz.max()) + 1 - int(ylens[b]) * 256)
print("last nonzero sample beyond len*256, per item:", offs)
