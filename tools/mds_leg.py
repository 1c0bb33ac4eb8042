"""The multi_device_synth leg of bench.py on its own (256 requests through MultiDeviceSynth on the visible devices):
   python tools/mds_leg.py [n_devices]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa
import bench
from vosk_tts_amd import weights as W
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
all_len = np.random.default_rng(1234).integers(20, 201, size=256)
print(bench.multi_device_synth_leg(W.default_hparams(), all_len, n, reps=5))
