"""Timeline of the persistent duration-predictor kernel (csrc/persist.hip.h): per step, over the busy workers, when the step was
entered, when its input cells had arrived, when it ended (cycles of the shader clock relative to the kernel's first stamp).
    VITS_PS_TRACE=/tmp/ps.bin python tools/ps_trace.py [T]      (eager stage call; the trace file is rewritten by every forward)"""
import os, sys, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
path = os.environ.setdefault("VITS_PS_TRACE", "/tmp/ps.bin")
import torch  # noqa
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib
T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
lib = VitsLib()
m = lib.create(W.synthetic_blob(W.default_hparams(), 1234), 0)
rng = np.random.default_rng(0)
x = rng.standard_normal((1, 192, T)).astype(np.float32)
for _ in range(3):
    m.duration(x, np.array([T], np.int64), np.array([2], np.int64), rng.standard_normal((1, 2, T)).astype(np.float32), 0.8)
raw = open(path, "rb").read()
P, MS, n, Tx = struct.unpack("4i", raw[:16])
kinds = struct.unpack(f"{n}i", raw[16:16 + 4 * n])
st = np.frombuffer(raw[16 + 4 * n:], dtype=np.int64).reshape(P, MS, 4)[:, :n]
t0 = st[st > 0].min()
names = {0: "PRE", 1: "COL", 2: "MM "}
print(f"T={Tx} workers={P} steps={n}; cycles relative to the first stamp (2.4 GHz: 2400 cycles = 1 us)")
prev_end = 0
for s in range(n):
    busy = st[:, s, 0] > 0
    a = st[busy, s]
    ent, got, end = a[:, 0] - t0, a[:, 1] - t0, a[:, 3] - t0
    print(f"step {s:2d} {names[kinds[s]]} workers {busy.sum():3d}  enter {ent.min():7d}..{ent.max():7d}  data {got.min():7d}..{got.max():7d}  end {end.min():7d}..{end.max():7d}"
          f"   wait {np.median(got - ent):6.0f}  work {np.median(end - got):6.0f}   step span {end.max() - prev_end:6d}")
    prev_end = end.max()
print(f"total {prev_end} cycles = {prev_end / 2400:.1f} us")
