"""Timeline of a persistent step program (csrc/persist.hip.h): per step, over the busy workers, how long they waited for their
input cells and how long they worked (cycles of the shader clock; stamps of different XCDs are not synchronised, so only
per-worker differences and the stamps of ONE worker are meaningful).
    python tools/ps_trace.py dp|enc|flow [T]      (eager stage call; the trace file is rewritten by every forward)"""
import os, sys, struct
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prog = sys.argv[1] if len(sys.argv) > 1 else "dp"
T = int(sys.argv[2]) if len(sys.argv) > 2 else (150 if prog == "flow" else 50)
path = os.environ.setdefault("VITS_PS_TRACE", "/tmp/ps.bin")
os.environ["VITS_PS_TRACE_PROG"] = prog + ".persist"
import torch  # noqa
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib
# the product library carries no stamps: the tracing build (make -C vosk_tts_amd/csrc libvits_mi355_pstrace.so)
lib = VitsLib(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vosk_tts_amd", "csrc", "libvits_mi355_pstrace.so"))
m = lib.create(W.synthetic_blob(W.default_hparams(), 1234), 0)
rng = np.random.default_rng(0)
for _ in range(3):
    if prog == "dp":
        m.duration(rng.standard_normal((1, 192, T)).astype(np.float32), np.array([T], np.int64), np.array([2], np.int64), rng.standard_normal((1, 2, T)).astype(np.float32), 0.8)
    elif prog == "enc":
        m.text_encoder(rng.integers(1, 62, size=(1, T)).astype(np.int64), np.array([T], np.int64), np.array([2], np.int64))
    else:
        m.flow(rng.standard_normal((1, 192, T)).astype(np.float32), np.array([T], np.int64), np.array([2], np.int64))
raw = open(path, "rb").read()
P, MS, n, Tx = struct.unpack("4i", raw[:16])
kinds = struct.unpack(f"{n}i", raw[16:16 + 4 * n])
st = np.frombuffer(raw[16 + 4 * n:], dtype=np.int64).reshape(P, MS, 8)[:, :n]
names = {0: "IDLE", 1: "MM", 2: "DDS", 3: "LN", 4: "EMB", 5: "ATT", 6: "MERGE", 7: "COUPLE", 8: "DUR", 9: "EXPAND", 10: "DDSL"}
print(f"{prog}: T={Tx} workers={P} steps={n}; 2400 cycles = 1 us")
# the timeline of ONE worker that is busy in most steps: rank 0
r0 = st[0]
t0 = r0[r0 > 0].min()
prev = 0
tot_wait = tot_work = 0
for s in range(n):
    busy = st[:, s, 0] > 0
    a = st[busy, s]
    got = np.where(a[:, 1] > 0, a[:, 1], a[:, 0])
    wait = np.median(got - a[:, 0]); work = np.median(a[:, 3] - got)
    e0 = r0[s, 3] - t0 if r0[s, 3] > 0 else -1
    print(f"step {s:2d} {names[kinds[s]]:6s} workers {busy.sum():3d}  wait {wait:6.0f} (max {np.max(got - a[:, 0]):6d})  work {work:6.0f} (max {np.max(a[:, 3] - got):6d})   rank0 end {e0:8d} (+{e0 - prev if e0 >= 0 else 0:6d})")
    if os.environ.get("PS_DETAIL") and s + 1 < n:
        both = busy & (st[:, s + 1, 0] > 0)
        if both.any():
            gap = st[both, s + 1, 0] - st[both, s, 3]
            print(f"        end of this step -> start stamp of the next (same worker): median {np.median(gap):.0f} cycles (p90 {np.sort(gap)[int(.9 * len(gap))]:.0f})")
    if os.environ.get("PS_DETAIL"):
        w = np.sort(a[:, 3] - got); g = np.sort(got - a[:, 0])
        q = lambda v, f: int(v[min(len(v) - 1, int(f * len(v)))])
        if kinds[s] == 5 and a[:, 5].max() > 0:
            ok = a[:, 5] > 0
            b = a[ok]; g2 = got[ok]
            md = lambda x: float(np.median(x)); mx = lambda x: float(np.max(x))
            print(f"        ATT phases median (max): poll done->tiles in LDS {md(b[:, 2] - g2):.0f} ({mx(b[:, 2] - g2):.0f}) | scores {md(b[:, 4] - b[:, 2]):.0f} ({mx(b[:, 4] - b[:, 2]):.0f}) | softmax {md(b[:, 5] - b[:, 4]):.0f} ({mx(b[:, 5] - b[:, 4]):.0f}) | PV + stores {md(b[:, 3] - b[:, 5]):.0f} ({mx(b[:, 3] - b[:, 5]):.0f})")
        if kinds[s] == 5 and os.environ.get("PS_ATT_ITEMS") and s == int(os.environ["PS_ATT_ITEMS"]):
            full = st[:, s]
            print("        PV+stores by worker:", [int(full[r, 3] - full[r, 5]) if full[r, 5] > 0 else -1 for r in range(min(P, 120))])
        if a[:, 6].max() > 0:
            md = lambda x: float(np.median(x))
            print(f"        MM phases (median cycles): start->poll done {md(got - a[:, 0]):.0f} | tile+barrier {md(a[:, 2] - got):.0f} | to MFMA loop {md(a[:, 7] - a[:, 2]):.0f} | MFMA loop {md(a[:, 4] - a[:, 7]):.0f} | res poll + partial tiles + barrier {md(a[:, 6] - a[:, 4]):.0f} | epilogue + stores {md(a[:, 3] - a[:, 6]):.0f}")
        print(f"        work p10/p50/p90/max {q(w, .1)}/{q(w, .5)}/{q(w, .9)}/{int(w[-1])}   wait p10/p50/p90 {q(g, .1)}/{q(g, .5)}/{q(g, .9)}   phases(median): poll->tile {np.median(a[:, 2] - got) if a[:, 2].max() > 0 else -1:.0f}")
    if e0 >= 0: prev = e0
    tot_wait += wait; tot_work += work
print(f"sum of medians: wait {tot_wait:.0f} work {tot_work:.0f} cycles = {(tot_wait + tot_work) / 2400:.1f} us; rank 0 timeline {prev / 2400:.1f} us")
