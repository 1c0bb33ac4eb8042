"""Round quantisation of the 128x128 conv kernel: 768 workgroup slots (256 CUs x 3); time vs number of tiles."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vosk_tts_amd.capi import VitsLib, op_conv1d
lib = VitsLib()
rng = np.random.default_rng(0)
for (B, T) in [(32, 9216), (32, 9600), (32, 12288), (32, 6144), (48, 6144), (32, 3072)]:
    for K, dil in ((3, 1), (11, 5)):
        x = rng.standard_normal((B, 128, T)).astype(np.float32); w = rng.standard_normal((128, 128, K)).astype(np.float32)
        op_conv1d(lib, x, w, np.zeros(128, np.float32), dil, 0.1)
