#!/usr/bin/env python3
"""Steady-state TFLOP/s of single conv launches on chosen shapes through vits_op_conv1d (timing build optional):
   python tools/convsweep.py "B,Cin,Cout,T,K,dil;..."  [env: VITS_KS_THRESHOLD / VITS_CONV_WP / VITS_BIG_BLOCKS select the kernel]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vosk_tts_amd.capi import VitsLib, op_conv1d
os.environ.setdefault("VITS_CONV_DBG", "30")
lib = VitsLib()
rng = np.random.default_rng(0)
for spec in sys.argv[1].split(";"):
    B, Cin, Cout, T, K, dil = (int(v) for v in spec.split(","))
    x = rng.standard_normal((B, Cin, T)).astype(np.float32); w = rng.standard_normal((Cout, Cin, K)).astype(np.float32)
    op_conv1d(lib, x, w, np.zeros(Cout, np.float32), dil, 0.1)
