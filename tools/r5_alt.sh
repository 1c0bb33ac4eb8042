#!/bin/bash
# the documented A/B switches must keep giving correct results: the parity subset under each
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
K="ragged or c3_full or padded_batch or tiny_b3 or mid_size or stages or c2_single or plain_generator or solo"
for e in "VITS_SP=2" "VITS_SP=0" "VITS_RAG_UNIFORM=1" "VITS_LN_SMALL=0" "VITS_NO_PERSIST=1"; do
  echo "== $e"; env $e timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --timeout 600 -k "$K" 2>&1 | tail -1
done
python bench.py --workload c2 --no-cpu-baseline --no-host-api --no-extras --no-batch32 --steps 10 2>/dev/null | head -c 200; echo
