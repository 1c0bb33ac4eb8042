#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "persistent or fast_path or coalesced or two_replicas or driver_timed or bert_conditioned" 2>&1 | tail -3
