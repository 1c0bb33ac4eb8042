#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "multi_device or solo or fast_path or pcm16 or model_synth or reentrant" > $O/r2_t6.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2_t6.log
tail -25 $O/r2_t6.log
