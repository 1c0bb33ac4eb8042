"""Request-level parallelism: N host threads, each calling stts_synthesize on its own pooled session / HIP stream
(the gRPC server's model: one Synth shared by a thread pool, server/tts_server.py:39-40,57).  Aggregate throughput."""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from vosk_tts_amd import weights as W, weights_stts as S
from vosk_tts_amd.capi import VitsLib
from vosk_tts_amd.capi_stts import SttsModel

lib = VitsLib()
voc = lib.create(W.synthetic_blob(W.hifigan_v1_vocoder_hparams(), 1234), 0)
m = SttsModel(lib, S.synthetic_blob(S.default_hparams(62, 5), 1234), voc)
rng = np.random.default_rng(0)
Tx = 50
ids = rng.integers(1, 62, size=(5, Tx)).astype(np.int64)
pde = np.full(Tx, 3.0, np.float32)
sc = np.array([0.8, 1.0, 0.8], np.float32)
for _ in range(3):
    m.synthesize(ids, sc, 2, None, pde, seed=1, want_mel=False)
for nthreads in (1, 2, 4, 8, 16, 32):
    per = 12
    def work(k):
        for i in range(per):
            m.synthesize(ids, sc, 2, None, pde, seed=k * 100 + i, want_mel=False)
    ths = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
    for k in range(nthreads): work  # noqa
    # warm the session pool for this concurrency level
    ws = [threading.Thread(target=lambda: m.synthesize(ids, sc, 2, None, pde, seed=5, want_mel=False)) for _ in range(nthreads)]
    [t.start() for t in ws]; [t.join() for t in ws]
    t0 = time.perf_counter()
    [t.start() for t in ths]; [t.join() for t in ths]
    dt = time.perf_counter() - t0
    n = nthreads * per
    sec = n * 150 * 256 / 22050
    print(f"{nthreads:3d} threads: {n} utterances in {dt*1e3:8.1f} ms -> {dt/n*1e3:6.2f} ms/utt, {sec/dt:8.1f}x real-time aggregate")
