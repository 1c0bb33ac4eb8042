#!/usr/bin/env python3
"""Per-(op, kernel) launch counts and average device time of one forward (HIP events around every launch, eager).
    python tools/profile_ops.py [c2|c3|c5] """
import os, sys
import numpy as np
import torch  # noqa: F401  (before the engine: one HIP runtime per process)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from vosk_tts_amd import weights as W  # noqa: E402
from vosk_tts_amd.capi import VitsLib, VitsDeviceSession  # noqa: E402

w = sys.argv[1] if len(sys.argv) > 1 else "c2"
free = len(sys.argv) > 2 and sys.argv[2] == "free"  # free-running durations: the duration predictor runs too
hp = W.default_hparams()
model = VitsLib().create(W.synthetic_blob(hp, 1234), 0)
ids, lengths, dur = bench.make_workload(w, np.random.default_rng(1234), 0, 1)
B, Tx = ids.shape; Ty = int(dur.sum(1).max()) * int(os.environ.get("PROFILE_TY_MULT", "1")); S = Ty * hp.hop_length
dev = torch.device("cuda", 0)
d = [torch.from_numpy(a).to(dev) for a in (ids, lengths, dur)]
sid = torch.full((B,), 2, dtype=torch.int64, device=dev)
audio = torch.empty((B, S), dtype=torch.float32, device=dev)
sess = VitsDeviceSession(model, B, Tx, Ty)
scales = np.array([0.8, 1.0, 0.8], np.float32)
def step():
    sess.synthesize_device(d[0].data_ptr(), d[1].data_ptr(), B, Tx, scales, sid.data_ptr(), 0 if free else d[2].data_ptr(), Ty, 7, audio.data_ptr(), S)
for _ in range(3): step()
sess.sync()
import time
t0 = time.perf_counter()
for _ in range(20): step()
sess.sync(); print(f"{w}: graph replay {(time.perf_counter()-t0)/20*1e3:.3f} ms/forward")
sess.set_options(use_graph=False, profile=True)
n = 5
for _ in range(n): step()
rep = sess.profile_report()
tot = sum(v[1] for v in rep.values()) / n
print(f"sum of kernel times {tot:.3f} ms, {sum(v[0] for v in rep.values())//n} launches")
for (op, k), v in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print(f"{op:16s} {k:38s} n={v[0]//n:4d} {v[1]/n*1e3:8.1f} us  avg {v[1]/v[0]*1e3:7.2f} us  {v[2]/max(v[1],1e-9)/1e9:7.2f} TF")
