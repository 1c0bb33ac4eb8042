#!/bin/bash
# mid-size conv shapes of the c3 batch through vits_op_conv1d under each kernel selection (plain library: steady-state time; timing build: phases of block 0)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-mid}; mkdir -p $O; cd $R
SH="1,192,768,3520,3,1;1,768,192,3520,3,1;1,192,576,3520,1,1;1,192,192,3520,1,1;1,192,192,10560,5,1;1,192,576,10560,1,1;1,192,384,10560,5,1;1,192,192,10560,1,1;1,768,96,10560,1,1;1,96,192,10560,1,1"
run() { # label env...
  local label=$1; shift
  env "$@" VITS_CONV_DBG=30 timeout 300 python tools/convsweep.py "$SH" 2>&1 | grep "conv dbg" | sed -e 's/: last launch.*back-to-back launches = / -> /' -e 's/; block 0.*//' -e "s/^/$label /"
}
{
run default A=1
run ks VITS_KS_THRESHOLD=100000000
run wp VITS_CONV_WP=2
run big128 VITS_BIG_BLOCKS=1
} > $O/sweep.txt 2>&1
cat $O/sweep.txt
VITS_CONV_DBG=30 CONVDBG_LIB=$R/vosk_tts_amd/csrc/libvits_mi355_timing.so timeout 300 python - > $O/phases.txt 2>&1 <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from vosk_tts_amd.capi import VitsLib, op_conv1d
lib = VitsLib(os.environ["CONVDBG_LIB"])
rng = np.random.default_rng(0)
for (B, Cin, Cout, T, K, dil) in [(1,192,768,3520,3,1), (1,192,192,10560,5,1), (1,192,384,10560,5,1), (1,768,96,10560,1,1)]:
    x = rng.standard_normal((B,Cin,T)).astype(np.float32); w = rng.standard_normal((Cout,Cin,K)).astype(np.float32)
    op_conv1d(lib, x, w, np.zeros(Cout, np.float32), dil, 0.1)
PY
grep -v "^blk" $O/phases.txt | head -60
