#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "VITS_KS_WAVES=4 VITS_CONV_LS=1" "VITS_KS_WAVES=16 VITS_CONV_LS=1" "VITS_CONV_LS=2" "VITS_CONV_LS=1 VITS_KS_WAVES=16 VITS_KS_SHAPE=12"; do
  echo "=== $cfg"
  env $cfg VITS_CONV_DBG=20 python tools/convdbg.py decoder 2>&1 | grep -E "conv dbg|wave" | cut -c1-260
done
