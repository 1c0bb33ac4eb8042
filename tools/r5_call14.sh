#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sp5; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 600 -k "software_pipelined" > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
bash tools/r5_ab.sh sp5 "VITS_SP=0" "VITS_SP=1" "VITS_SP=1 VITS_SP_MAXBLK=2048" "VITS_SP=2"
