#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c2; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "rearms or starve or across_eligible or padded_batch_of_8 or c4_shard or coalesced" > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -5 $O/tests.log
timeout 600 python tools/split_probe.py c3 1 2 4 > $O/split_c3.txt 2>&1; cat $O/split_c3.txt | grep stream
timeout 600 python tools/split_probe.py s16 1 2 4 > $O/split_s16.txt 2>&1; cat $O/split_s16.txt | grep stream
