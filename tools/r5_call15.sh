#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/co1; mkdir -p $O; cd $R
timeout 600 python tools/coalesce_probe.py 16 0.5 > $O/probe16.txt 2>&1; grep inflight $O/probe16.txt
timeout 600 python tools/coalesce_probe.py 4 0.4 > $O/probe4.txt 2>&1; grep inflight $O/probe4.txt
timeout 600 python tools/coalesce_probe.py 1 0.3 > $O/probe1.txt 2>&1; grep inflight $O/probe1.txt | head -4
