"""Shader clock INSIDE one grouped ResBlock launch of a real c3 forward (timing build): s_memtime cycles of every workgroup's life over its
wall-clock life (block trace words of conv_mfma_kernel, CONV_TIMING).     VITS_DBG_GROUPED=<n> python tools/clock_in_forward.py [c3|s16]"""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CIF_CHILD"):
    sys.path.insert(0, root)
    import numpy as np
    import torch  # noqa
    import importlib.util
    from vosk_tts_amd import weights as W
    from vosk_tts_amd.capi import VitsLib
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py")); bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    lib = VitsLib(os.path.join(root, "vosk_tts_amd", "csrc", "libvits_mi355_timing.so"))
    hp = W.default_hparams()
    m = lib.create(W.synthetic_blob(hp, 1234), 0)
    ids, lengths, dur = bench.make_workload(os.environ["CIF_CHILD"], np.random.default_rng(1234))
    B = ids.shape[0]
    for _ in range(4):
        m.synthesize(ids, lengths, np.array([0.8, 1.0, 0.8], np.float32), np.full(B, 2, np.int64), forced_durations=dur, seed=1)
    sys.exit(0)
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
env = dict(os.environ, CIF_CHILD=wl, VITS_NO_FASTPATH="1")
env.setdefault("VITS_DBG_GROUPED", "20")
r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
blk = []
for l in r.stderr.splitlines():
    if l.startswith("[in-forward conv dbg]"): print(l)
    if l.startswith("blk "):
        _, i, s, e, hw, xcc = l.split(); blk.append((int(i), int(s), int(e), int(hw), int(xcc)))
if not blk:
    print(r.stderr[-2000:]); sys.exit(1)
clk = sorted((x >> 8) / ((e - s) * 10.0) for _, s, e, _, x in blk if (x >> 8) and e > s)
dur = sorted((e - s) / 100 for _, s, e, _, x in blk if e > s)
print(f"{wl}: grouped launch #{env['VITS_DBG_GROUPED']}: {len(blk)} workgroups traced, makespan {max(b[2] for b in blk) / 100:.1f} us, workgroup life p50 {dur[len(dur)//2]:.1f} us")
if clk: print(f"   shader clock over a workgroup's life (s_memtime cycles / wall time), GHz: p10 {clk[len(clk)//10]:.3f} p50 {clk[len(clk)//2]:.3f} p90 {clk[len(clk)*9//10]:.3f}  ({len(clk)} workgroups)")
