#!/usr/bin/env python3
"""Does a 32-item batch run faster as n independent sub-batches on n streams (n device sessions, each with its own graph) than as one
batch?  The mid-size launches of the acoustic half (text encoder, duration predictor, flow) underfill the chip at c3 size; concurrent
streams fill each other's idle CUs.  Uses only the existing C ABI (one VitsDeviceSession per sub-batch).
    python tools/split_probe.py [c3|c4|s16] [n ...]"""
import os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vosk_tts_amd import weights as W  # noqa: E402
from vosk_tts_amd.capi import VitsLib, VitsDeviceSession  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "c3"
ns = [int(a) for a in sys.argv[2:]] or [1, 2, 4]
hp = W.default_hparams()
model = VitsLib().create(W.synthetic_blob(hp, 1234), 0)
ids, lengths, dur = bench.make_workload(w, np.random.default_rng(1234), 0, 1)
B = ids.shape[0]
dev = torch.device("cuda", 0)
scales = np.array([0.8, 1.0, 0.8], np.float32)
order_modes = {"as_is": np.arange(B), "sorted": np.argsort(-lengths), "dealt": None}
for n in ns:
    for mode in (["as_is"] if n == 1 else ["as_is", "sorted", "dealt"]):
        if mode == "dealt":  # longest-first dealt round robin: every part gets the same length mix
            o = np.argsort(-lengths)
            parts = [o[k::n] for k in range(n)]
        else:
            o = order_modes[mode]
            parts = np.array_split(o, n)
        subs = []
        for p in parts:
            li = lengths[p]; Tx = int(li.max()); di = dur[p][:, :Tx]; Ty = int(di.sum(1).max())
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            d_ids, d_len, d_dur = t(ids[p][:, :Tx]), t(li), t(di)
            d_sid = torch.full((len(p),), 2, dtype=torch.int64, device=dev)
            d_audio = torch.empty((len(p), Ty * 256), dtype=torch.float32, device=dev)
            s = VitsDeviceSession(model, len(p), Tx, Ty)
            s.set_options(use_graph=True, profile=False); s.set_sdp_always(True)
            subs.append((s, d_ids, d_len, d_sid, d_dur, d_audio, len(p), Tx, Ty))
        def step():
            for (s, d_ids, d_len, d_sid, d_dur, d_audio, b, Tx, Ty) in subs:
                s.synthesize_device(d_ids.data_ptr(), d_len.data_ptr(), b, Tx, scales, d_sid.data_ptr(), d_dur.data_ptr(), Ty, 7, d_audio.data_ptr(), Ty * 256)
        def sync():
            for sub in subs: sub[0].sync()
        for _ in range(3): step()
        sync()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(10): step()
            sync()
            ts.append((time.perf_counter() - t0) / 10 * 1e3)
        print(f"{w}: {n} stream(s) [{mode}]: {np.median(ts):.3f} ms per {B}-item batch (min {min(ts):.3f})", flush=True)
        for sub in subs: sub[0].close()
