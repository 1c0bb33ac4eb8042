#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/rag1; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 600 -k "ragged or c3_full or c4_shard or padded_batch or plain_generator or driver_timed or solo or bf16x3 or stts_batch or multi_device or mid_size or software_pipelined" > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -6 $O/tests.log
WL="c3 c4 s8" bash tools/r5_ab.sh rag1 "VITS_RAG_UNIFORM=1" "VITS_RAG_UNIFORM=0"
