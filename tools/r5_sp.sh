#!/bin/bash
# conv_sp iteration: parity test, sweep sp off / all, block trace at 1 wg/CU
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-sp}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 600 -k "software_pipelined" > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
SH="1,192,768,3520,3,1;1,768,192,3520,3,1;1,192,576,3520,1,1;1,192,192,3520,1,1;1,192,192,10560,5,1;1,192,576,10560,1,1;1,192,384,10560,5,1;1,192,192,10560,1,1;1,768,64,10560,1,1;8,256,256,564,3,1;8,256,256,564,11,5;8,128,128,2256,7,3"
run() { local label=$1; shift
  env "$@" VITS_CONV_DBG=30 timeout 300 python tools/convsweep.py "$SH" 2>&1 | grep "conv dbg" | sed -e 's/: last launch.*back-to-back launches = / -> /' -e 's/; block 0.*//' -e "s/^/$label /"; }
{ run sp_off VITS_SP=0; run sp_all VITS_SP=2; } > $O/sweep.txt 2>&1
paste -d'|' <(grep sp_off $O/sweep.txt) <(grep sp_all $O/sweep.txt | sed 's/.*->/->/')
VITS_SP=2 timeout 600 python tools/bt_conv.py "8,256,256,564,3,1;1,768,192,3520,3,1" > $O/bt_sp.txt 2>&1
grep -v "^\[conv" $O/bt_sp.txt
