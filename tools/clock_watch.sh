#!/bin/bash
# shader clock / socket power while a workload runs: tools/clock_watch.sh "<command>" [samples]   (rocm-smi polled every ~0.2 s)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash -c "$1" > /dev/null 2>&1 &
PID=$!
sleep ${DELAY:-8}
for i in $(seq 1 ${2:-12}); do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|fclk|mclk" | tr -s ' ' | tr '\n' ';'; echo
  sleep 0.2
done
wait $PID
