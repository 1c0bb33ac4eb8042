#!/bin/bash
# call 1: new robustness tests + per-shape c3 / s8 breakdown + baseline bench lines
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/c1; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "rearms or starve or across_eligible or padded_batch_of_8 or c3_full_size or c4_shard or coalesced" > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -15 $O/tests.log
VITS_PROF_SHAPES=1 timeout 300 python tools/profile_ops.py c3 > $O/c3_shapes.txt 2>&1; echo "c3 rc=$?"
VITS_PROF_SHAPES=1 timeout 300 python tools/profile_ops.py s8 > $O/s8_shapes.txt 2>&1; echo "s8 rc=$?"
timeout 300 python tools/profile_ops.py c2 > $O/c2_ops.txt 2>&1
timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-host-api > $O/c3_bench.json.txt 2>$O/c3_bench.err; echo "bench c3 rc=$?"
head -3 $O/c3_shapes.txt
