"""B=1 decoder ResBlock convs through vits_op_conv1d with the K-split tile shape forced by VITS_KS_SHAPE
(run once per shape: VITS_CONV_DBG=20 VITS_KS_SHAPE=12 python tools/ks_shapes.py; the 64x32 / 64x64 tiles this
was also used for were measured slower for B=1 -- DESIGN.md §6 -- and their instantiations removed again)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vosk_tts_amd.capi import VitsLib, op_conv1d
lib = VitsLib()
rng = np.random.default_rng(0)
for (C, T) in [(256, 600), (128, 2400)]:
    for (K, dil) in [(3, 1), (7, 3), (11, 5)]:
        x = rng.standard_normal((1, C, T)).astype(np.float32); w = rng.standard_normal((C, C, K)).astype(np.float32)
        op_conv1d(lib, x, w, np.zeros(C, np.float32), dil, 0.1)
