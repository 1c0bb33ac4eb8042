#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for cfg in "VITS_CONV_WP=2 CONVDBG_SLOPE=1.0" "VITS_CONV_WP=2 CONVDBG_SLOPE=0.1"; do
  echo "=== $cfg"
  env $cfg VITS_CONV_DBG=20 timeout 300 python tools/convdbg.py decoder 2>&1 | grep -E "conv dbg|wave" | cut -c1-200 | grep -A9 -E "T=2400 K=11"
done
