#!/bin/bash
# Round-4 measurement call (one lease): persistent-path parity subset, A/B of the persistent-program switches on the c2 bench,
# stage timelines, quick FETCH_SIZE comparison.  Usage: gpurun -- 'bash tools/gpu_lease.sh [tests] [ab] [trace] [pmc]'
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
c2() {  # one c2-only bench line -> "ms_per_step {persist ops}"
  BENCH_SKIP_FINITE_CHECK=1 timeout 300 python bench.py --no-batch32 --no-cpu-baseline --no-host-api --no-extras --steps 50 2>$O/bench_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], d['launches_per_forward'], {k:round(v,4) for k,v in r['by_op_ms_per_forward'].items()})"
}
for what in "$@"; do case $what in
tests)
  timeout 900 python -m pytest tests -m gpu -q -x -k "${TESTS_K:-persistent or driver_timed or timeout or fast_path or c2_single}" 2>&1 | tail -5 | tee $O/tests.txt;;
alltests)
  timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -8 | tee $O/alltests.txt;;
ab)
  for cfg in ${AB_CFGS:-"0:0" "1:0" "1:1" "0:1" "1:1"}; do
    x=${cfg%%:*}; t=${cfg##*:}
    echo "VITS_PS_XCD=$x VITS_PS_TUNE=$t: $(VITS_PS_XCD=$x VITS_PS_TUNE=$t c2)" | tee -a $O/ab.txt
  done;;
libs)  # A/B of alternative builds (tools/ab_build.sh): every tools/bt/bt_*.so, twice each, interleaved
  for rep in 1 2; do for so in tools/bt/bt_*.so; do for tune in ${LIB_TUNES:-0}; do
    echo "$so VITS_PS_TUNE=$tune: $(VITS_MI355_LIB=$R/$so VITS_PS_TUNE=$tune c2)" | tee -a $O/libs.txt
  done; done; done;;
bench)  # the default driver line
  timeout 900 python bench.py > $O/default_bench.json.txt 2> $O/default_bench.err; echo "bench rc=$?"; tail -c 300 $O/default_bench.err
  python tools/bench_summary.py $O/default_bench.json.txt;;
c2ab)  # c2 under environment variants: C2_ENVS="A=1 B=2|A=3" (| separates variants), each twice, interleaved
  IFS='|' read -ra VARS <<< "${C2_ENVS:-}"
  for rep in 1 2; do for v in "" "${VARS[@]}"; do
    echo "c2 [$v]: $(env $v bash -c "$(declare -f c2); R=$R; O=$O; c2")" | tee -a $O/c2ab.txt
  done; done;;
c3ab)  # c3 (batch 32) under environment variants: C3_ENVS="A=1 B=2|A=3" (| separates variants)
  IFS='|' read -ra VARS <<< "${C3_ENVS:-}"
  for v in "" "${VARS[@]}"; do
    echo "c3 [$v]: $(env $v timeout 300 python bench.py --workload c3 --no-cpu-baseline --no-host-api --no-extras 2>$O/c3_err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], r['kernel'], r['frac'], {k:round(v,3) for k,v in list(r['by_kernel_ms_per_forward'].items())[:9]})")" | tee -a $O/c3ab.txt
  done;;
trace)
  for p in enc dp flow; do PS_DETAIL=1 timeout 200 python tools/ps_trace.py $p > $O/trace_$p.txt 2>&1; tail -2 $O/trace_$p.txt; done;;
pmc)
  cd /tmp && export TMPDIR=/tmp
  for x in 0 1; do
    rm -rf $O/pmcq$x
    VITS_PS_XCD=$x timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmcq$x -o f -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch32 --no-host-api --no-extras --min-seconds 0 --no-graph > /dev/null 2>&1
    python - <<PY | tee -a $O/pmcq.txt
import csv,glob
v=[float(r["Counter_Value"]) for p in glob.glob("$O/pmcq$x/**/*counter_collection.csv",recursive=True) for r in csv.DictReader(open(p)) if r["Kernel_Name"].startswith("persist_kernel")]
print("VITS_PS_XCD=$x persist_kernel FETCH_SIZE KB per launch (x2 = bytes):", round(sum(v)/max(len(v),1),1), "launches", len(v))
PY
    rm -rf $O/pmcq$x
  done
  cd $R;;
esac; done
