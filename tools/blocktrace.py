"""Block trace of one grouped decoder launch inside a c2 forward (timing build, tools/bt/bt_trace.so):
   VITS_DBG_GROUPED=<n> python tools/blocktrace.py      prints per-CU occupancy and the launch's makespan (wall clock, 10 ns units)"""
import os, sys, subprocess, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("BT_CHILD"):
    sys.path.insert(0, root)
    import numpy as np
    from vosk_tts_amd import weights as W
    from vosk_tts_amd.capi import VitsLib
    lib = VitsLib(os.path.join(root, "tools", "bt", "bt_trace.so"))
    hp = W.default_hparams()
    m = lib.create(W.synthetic_blob(hp, 1234), 0)
    rng = np.random.default_rng(0)
    ids = rng.integers(1, hp.n_vocab, size=(1, 50)).astype(np.int64)
    fd = np.full((1, 50), 3, np.int32)
    for _ in range(6):
        m.synthesize(ids, np.array([50]), np.array([0.667, 1.0, 0.8], np.float32), np.array([0]), forced_durations=fd, seed=1)
    sys.exit(0)
env = dict(os.environ, BT_CHILD="1", VITS_NO_FASTPATH="1")
r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
blk = []
for l in r.stderr.splitlines():
    if l.startswith("[in-forward conv dbg]") or l.startswith("   wave"): print(l)
    if l.startswith("blk "):
        _, i, s, e, hw, xcc = l.split(); blk.append((int(i), int(s), int(e), int(hw), int(xcc)))
if not blk:
    print(r.stderr[-2000:]); sys.exit(1)
end = max(b[2] for b in blk)
print(f"{len(blk)} workgroups; makespan {end / 100:.2f} us (first start -> last end)")
# hardware id: HW_ID bits: wave 0-3, simd 4-5, pipe 6-7, cu 8-11, sh 12, se 13-15 (gfx9); XCC_ID reg low bits
def cu_of(hw, xcc): return (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
by = collections.defaultdict(list)
for i, s, e, hw, xcc in blk: by[cu_of(hw, xcc)].append((s, e, i))
print(f"{len(by)} distinct CUs used")
durs = sorted(e - s for _, s, e, _, _ in blk)
print("workgroup duration us: min %.2f median %.2f max %.2f" % (durs[0] / 100, durs[len(durs) // 2] / 100, durs[-1] / 100))
starts = sorted(s for _, s, _, _, _ in blk)
print("start times us: median %.2f  90%% %.2f  last %.2f" % (starts[len(starts) // 2] / 100, starts[int(len(starts) * .9)] / 100, starts[-1] / 100))
n3 = len(blk) // 3
for g, name in enumerate(("heaviest", "middle", "lightest")):
    sub = [b for b in blk if g * n3 <= b[0] < (g + 1) * n3]
    if sub: print(f"group {name}: start median {sorted(b[1] for b in sub)[len(sub)//2] / 100:.2f} us, duration median {sorted(b[2]-b[1] for b in sub)[len(sub)//2] / 100:.2f} us, last end {max(b[2] for b in sub) / 100:.2f} us")
cnt = collections.Counter(len(v) for v in by.values()); print("workgroups per CU:", sorted(cnt.items()))
busiest = sorted(by.items(), key=lambda kv: -max(e for _, e, _ in kv[1]))[:6]
for cu, v in busiest: print(" CU", cu, [(i, round(s / 100, 2), round(e / 100, 2)) for s, e, i in sorted(v)])
