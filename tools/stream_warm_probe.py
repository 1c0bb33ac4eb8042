"""How many stream opens of one shape does it take until time-to-first-audio settles?  python tools/stream_warm_probe.py"""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
import importlib.util
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
lib = VitsLib(); hp = W.default_hparams(); model = lib.create(W.synthetic_blob(hp, 1234), 0)
ids, lengths, dur = bench.make_workload("c5", np.random.default_rng(1234))
sc = np.array([0.8, 1.0, 0.8], np.float32)
oneshot = "--oneshot" in sys.argv
for it in range(8):
    t0 = time.perf_counter()
    g = model.stream(ids[:1], sc, 2, chunk_frames=128, forced_durations=dur[:1], seed=7)
    first = next(g)
    t1 = time.perf_counter()
    n = len(first) + sum(len(c) for c in g)
    t2 = time.perf_counter()
    extra = ""
    if oneshot:
        c0 = time.perf_counter(); model.synthesize(ids[:1], lengths[:1], sc, [2], forced_durations=dur[:1], seed=7); extra = f"  one-shot {(time.perf_counter()-c0)*1e3:.2f} ms"
    print(f"open {it}: first audio {(t1-t0)*1e3:7.2f} ms, all chunks {(t2-t0)*1e3:7.2f} ms{extra}")
