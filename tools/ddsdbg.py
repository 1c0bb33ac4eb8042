"""Phase stamps of one DDSConv-layer launch of the small-tile conv kernel (timing build): VITS_DBG_DDS=<i> python tools/ddsdbg.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vosk_tts_amd import weights as W
from vosk_tts_amd.capi import VitsLib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = VitsLib(os.path.join(root, "vosk_tts_amd", "csrc", "libvits_mi355_timing.so"))
model = lib.create(W.synthetic_blob(W.default_hparams(), 1234), 0)
rng = np.random.default_rng(1)
x = rng.standard_normal((1, 192, 50)).astype(np.float32)
for _ in range(3):
    model.duration(x, [50], [2], rng.standard_normal((1, 2, 50)).astype(np.float32), 0.8)
