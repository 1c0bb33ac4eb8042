#!/usr/bin/env python3
"""Block trace of single conv launches (timing build, vits_op_conv1d): start spread, durations, per-CU counts, makespan.
    [env selecting the kernel] python tools/bt_conv.py "B,Cin,Cout,T,K,dil;..." """
import os, sys, subprocess, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("BT_CHILD"):
    sys.path.insert(0, root)
    import numpy as np
    import torch  # noqa
    from vosk_tts_amd.capi import VitsLib, op_conv1d
    lib = VitsLib(os.environ.get("BT_LIB", os.path.join(root, "vosk_tts_amd", "csrc", "libvits_mi355_timing.so")))
    rng = np.random.default_rng(0)
    B, Cin, Cout, T, K, dil = (int(v) for v in os.environ["BT_CHILD"].split(","))
    x = rng.standard_normal((B, Cin, T)).astype(np.float32); w = rng.standard_normal((Cout, Cin, K)).astype(np.float32)
    op_conv1d(lib, x, w, np.zeros(Cout, np.float32), dil, 0.1)
    sys.exit(0)
for spec in sys.argv[1].split(";"):
    env = dict(os.environ, BT_CHILD=spec, VITS_CONV_DBG="20", VITS_CONV_BT="1")
    r = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
    blk = []
    for l in r.stderr.splitlines():
        if l.startswith("[conv dbg]"): print(l.split("; block 0")[0])
        if "big-tile] wave 0" in l or l.startswith("   wave  0"): print(l)
        if l.startswith("blk "):
            _, i, s, e, hw, xcc = l.split(); blk.append((int(i), int(s), int(e), int(hw), int(xcc)))
    if not blk:
        print(r.stderr[-1500:]); continue
    end = max(b[2] for b in blk)
    def cu_of(hw, xcc): return (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
    clk = sorted((x >> 8) / ((e - s) * 10.0) for _, s, e, _, x in blk if (x >> 8) and e > s)  # shader-clock cycles / wall ns (big-tile kernel only)
    if clk: print(f"   shader clock over a workgroup's life (s_memtime cycles / wall time), GHz: p10 {clk[len(clk)//10]:.3f} p50 {clk[len(clk)//2]:.3f} p90 {clk[len(clk)*9//10]:.3f}")
    by = collections.defaultdict(list)
    for i, s, e, hw, xcc in blk: by[cu_of(hw, xcc)].append((s, e, i))
    durs = sorted(e - s for _, s, e, _, _ in blk)
    starts = sorted(s for _, s, _, _, _ in blk)
    q = lambda v, f: v[min(len(v) - 1, int(len(v) * f))] / 100
    print(f"   {len(blk)} workgroups on {len(by)} CUs; makespan {end / 100:.2f} us; wg duration min {q(durs,0):.2f} p50 {q(durs,.5):.2f} p90 {q(durs,.9):.2f} max {q(durs,1):.2f}; "
          f"start p50 {q(starts,.5):.2f} p90 {q(starts,.9):.2f} last {q(starts,1):.2f}")
    cnt = collections.Counter(len(v) for v in by.values()); print("   workgroups per CU:", sorted(cnt.items()))
    ends = sorted(max(e for _, e, _ in v) for v in by.values())
    print(f"   CU finish time us: p10 {q(ends,.1):.2f} p50 {q(ends,.5):.2f} p90 {q(ends,.9):.2f} max {q(ends,1):.2f}")
