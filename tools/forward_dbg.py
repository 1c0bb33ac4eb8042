"""Phase stamps of single conv launches INSIDE a real c2 forward (timing build of the library)."""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
import torch
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib
lib = VitsLib(os.path.join(%r, "vosk_tts_amd", "csrc", "libvits_mi355_timing.so"))
model = lib.create(W.synthetic_blob(W.default_hparams(), 1234), 0)
rng = np.random.default_rng(1234)
ids = rng.integers(1, 62, size=(1, 50)).astype(np.int64)
for _ in range(3):
    model.synthesize(ids, [50], [0.8, 1.0, 0.8], [2], forced_durations=np.full((1, 50), 3, np.int32), seed=1)
''' % (root, root)
# a forward with pinned durations issues ~103 conv launches; 3 forwards -> sample launches of the third one
for want in sys.argv[1:] or ["209", "212", "240", "300"]:
    env = dict(os.environ, VITS_DBG_LAUNCH=want)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("\n".join(l for l in r.stderr.splitlines() if "dbg" in l or "wave 0" in l))
