"""A/B of builds of the library on batch-sized single conv launches (3 full rounds of the big-tile kernel's 768 workgroup slots):
   BT_LIBS="a.so b.so" VITS_CONV_DBG=12 python tools/bigtile_ab.py      (each lib in its own process: the C ABI has one library per process)
Used in round 2 with builds that removed one class of work at a time (weight loads, staging, barrier; results garbage, time only)
to find what the 128x128 fp32 kernel's tap loop was paying for -- DESIGN.md section 6, "Big-tile kernel"."""
import sys, os, subprocess, numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, root)
    import torch  # noqa
    from vosk_tts_amd.capi import VitsLib, op_conv1d
    lib = VitsLib(sys.argv[1])
    rng = np.random.default_rng(0)
    for (B, C, T, K, dil) in [(32, 256, 4608, 3, 1), (32, 256, 4608, 11, 5), (32, 128, 9216, 3, 1)]:
        x = rng.standard_normal((B, C, T)).astype(np.float32); w = rng.standard_normal((C, C, K)).astype(np.float32)
        op_conv1d(lib, x, w, np.zeros(C, np.float32), dil, 0.1)
else:
    for so in os.environ["BT_LIBS"].split():
        r = subprocess.run([sys.executable, __file__, os.path.join(root, so)], capture_output=True, text=True, env=dict(os.environ, VITS_CONV_DBG=os.environ.get("VITS_CONV_DBG", "12")))
        print("==", so)
        for l in r.stderr.splitlines():
            if l.startswith("[conv dbg]"):
                print("  ", l.split(": last launch")[0].replace("[conv dbg] ", ""), "->", l.split("launches = ")[1].split(";")[0])
        if r.returncode: print("   rc", r.returncode, r.stderr[-300:])
