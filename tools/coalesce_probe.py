#!/usr/bin/env python3
"""Where the time of the 16-thread serving leg goes (bench.py host_api.concurrent): per engine call of the request coalescer its batch
size and duration, the fraction of the wall clock with >= 1 / >= 2 engine calls running, requests/s -- for several values of the
calls allowed in flight.    python tools/coalesce_probe.py [threads=16] [seconds=0.6]"""
import os, sys, tempfile, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from vosk_tts_amd import Model, Synth  # noqa: E402
from vosk_tts_amd import weights as W  # noqa: E402
from vosk_tts_amd.toymodel import PHONEMES, write_toy_model  # noqa: E402

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
hp = W.default_hparams(n_vocab=len(PHONEMES))
text = "привет мир привет мир."
with tempfile.TemporaryDirectory() as d:
    write_toy_model(d, hp)
    model = Model(model_path=d, device=0)
    synth = Synth(model)
    sess = model.onnx
    co = sess.coalescer
    inner = co._run_batch
    log, lock = [], threading.Lock()

    def timed(key, reqs):
        t0 = time.perf_counter()
        try:
            return inner(key, reqs)
        finally:
            t1 = time.perf_counter()
            with lock:
                log.append((t0, t1, len(reqs)))
    co._run_batch = timed

    def run(inflight, gather_us=0.0):
        co.max_inflight = inflight
        co.gather_us = gather_us
        g0 = co.gathered
        for _ in range(4):
            synth.synth_audio(text, speaker_id=2)
        for timed_leg in (False, True):
            stop = time.perf_counter() + (seconds if timed_leg else 0.5)
            cnt = [0] * n_threads
            del log[:]

            def worker(k):
                while time.perf_counter() < stop:
                    synth.synth_audio(text, speaker_id=2)
                    cnt[k] += 1
            t_all = time.perf_counter()
            th = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
            [t.start() for t in th]
            [t.join() for t in th]
            el = time.perf_counter() - t_all
        ev = sorted([(a, 1) for a, b, n in log] + [(b, -1) for a, b, n in log])
        busy1 = busy2 = 0.0
        depth, last = 0, t_all
        for t, dlt in ev:
            if depth >= 1: busy1 += t - last
            if depth >= 2: busy2 += t - last
            depth += dlt; last = t
        by = {}
        for a, b, n in log:
            by.setdefault(n, []).append(b - a)
        sizes = " ".join(f"{n}:{len(v)}x{np.median(v)*1e3:.2f}ms" for n, v in sorted(by.items()))
        print(f"inflight {inflight:2d} gather {gather_us:5.0f} us ({co.gathered - g0} waits): {sum(cnt)/el:7.1f} req/s  calls {len(log):4d}  mean batch {sum(n for _,_,n in log)/max(len(log),1):.2f}  "
              f">=1 call running {busy1/el:.2f}  >=2 {busy2/el:.2f}  | size:count x median call: {sizes}", flush=True)

    for k, g in ((8, 0.0), (1, 400.0), (2, 400.0), (2, 800.0), (3, 400.0), (4, 400.0), (8, 400.0)):
        run(k, g)
    sess.close()
