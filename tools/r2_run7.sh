#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "bert_conditioned_vits or model_synth or multi_device" > $O/r2_t7.log 2>&1; echo "pytest rc=$?" | tee -a $O/r2_t7.log
tail -25 $O/r2_t7.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
