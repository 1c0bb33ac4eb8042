import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vosk_tts_amd.capi import VitsLib, op_conv1d
lib = VitsLib()
rng = np.random.default_rng(0)
for (B, Cin, Cout, T, K, dil) in [(1,192,192,50,1,1), (1,768,192,50,3,1), (1,192,384,150,5,1), (1,256,256,600,11,5), (1,128,128,2400,7,1)]:
    x = rng.standard_normal((B,Cin,T)).astype(np.float32); w = rng.standard_normal((Cout,Cin,K)).astype(np.float32)
    op_conv1d(lib, x, w, np.zeros(Cout, np.float32), dil, 0.1)

print("--- big-tile regime", file=sys.stderr)
for (B, Cin, Cout, T, K, dil) in [(32,128,128,9600,3,1), (32,128,128,9600,7,3), (32,128,128,9600,11,5), (32,256,256,2400,7,1)]:
    x = rng.standard_normal((B,Cin,T)).astype(np.float32); w = rng.standard_normal((Cout,Cin,K)).astype(np.float32)
    op_conv1d(lib, x, w, np.zeros(Cout, np.float32), dil, 0.1)
