#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/rag2; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -k "ragged or c3_full or padded_batch or tiny_b3" > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/tests.log
WL="c3 c4 s8 s16" bash tools/r5_ab.sh rag2 "VITS_RAG_UNIFORM=0"
