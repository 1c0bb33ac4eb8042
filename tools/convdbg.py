"""Single conv launches through vits_op_conv1d with the timing build (phase stamps of block 0 + steady-state time per launch):
   VITS_CONV_DBG=20 [VITS_KS_WAVES=..] [VITS_CONV_WP=..] [CONVDBG_LIB=..] [CONVDBG_SLOPE=..] python tools/convdbg.py [decoder|small]"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from vosk_tts_amd.capi import VitsLib, op_conv1d
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tl = os.environ.get("CONVDBG_LIB", os.path.join(root, "vosk_tts_amd", "csrc", "libvits_mi355_timing.so"))
lib = VitsLib(tl if os.path.exists(tl) and not os.environ.get("CONVDBG_PLAIN") else None)
rng = np.random.default_rng(0)
which = sys.argv[1] if len(sys.argv) > 1 else "decoder"
shapes = {"decoder": [(1,256,256,600,3,1), (1,256,256,600,7,3), (1,256,256,600,11,5), (1,128,128,2400,3,1), (1,128,128,2400,11,5), (1,512,1024,150,4,1)],
          "small": [(1,192,192,50,1,1), (1,768,192,50,3,1), (1,192,384,150,5,1), (1,192,576,150,1,1)]}[which]
for (B, Cin, Cout, T, K, dil) in shapes:
    x = rng.standard_normal((B,Cin,T)).astype(np.float32); w = rng.standard_normal((Cout,Cin,K)).astype(np.float32)
    op_conv1d(lib, x, w, np.zeros(Cout, np.float32), dil, float(os.environ.get('CONVDBG_SLOPE', '0.1')))
