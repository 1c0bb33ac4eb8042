#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 900 -k "bf16x3" > $O/r2_t15.log 2>&1; echo "pytest rc=$?"
tail -12 $O/r2_t15.log | cut -c1-220
bash tools/r2_run16.sh
