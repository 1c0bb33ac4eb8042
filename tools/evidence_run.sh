#!/bin/bash
# evidence run of a round (set PROFILE_ROUND in bench.py accordingly): full GPU test suite, default bench line, rocprofv3 kernel stats (c2-only and c3), PMC passes (c2, c3).
# Outputs land in gpurun_out/final/ under the names profiles/ expects; copy them to profiles/ afterwards.
# EVIDENCE_HEAD=<git sha of the tree that is measured> is written to ${RD}_HEAD.txt (the GPU box has no .git): bench.py prints it beside
# every figure it reads from the committed profiles.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/final; mkdir -p $O
RD=${ROUND:-r6}   # prefix of the files (= PROFILE_ROUND in bench.py)
cd $R
echo "${EVIDENCE_HEAD:-unknown}" > $O/${RD}_HEAD.txt
bf16x3_profiles() {
  # the split-bf16 second line (c3, conv_precision = 1): its own kernel stats and counter passes
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/c3b -o trace -- python $R/bench.py --workload c3 --precision bf16x3 --no-cpu-baseline --no-host-api > $O/${RD}_c3_bf16x3_bench_under_rocprof.json.txt 2> $O/prof_c3b.err
  cp $(find $O/prof/c3b -name '*kernel_stats.csv' | head -1) $O/${RD}_c3_bf16x3_bench_rocprofv3_kernel_stats.csv
  rm -rf $O/prof
  cd $R
  timeout 900 bash tools/pmc_passes.sh c3 "--precision bf16x3" _bf16x3 > $O/pmc_c3_bf16x3.log 2>&1
  cp $R/gpurun_out/pmc_c3_bf16x3/pmc_c3_bf16x3.json $O/${RD}_pmc_c3_bf16x3.json
  rm -rf $R/gpurun_out/pmc_c3_bf16x3
}
if [ "$1" == "bf16x3_only" ]; then bf16x3_profiles; ls -la $O; exit 0; fi
if [ "$1" == "c2_profiles_only" ]; then  # the c2 kernel stats + counter passes again (e.g. after a change to the single-utterance path)
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/c2 -o trace -- python $R/bench.py --no-batch32 --no-cpu-baseline --no-host-api --no-extras --steps 50 > $O/${RD}_c2_only_bench_under_rocprof.json.txt 2> $O/prof_c2.err
  cp $(find $O/prof/c2 -name '*kernel_stats.csv' | head -1) $O/${RD}_c2_only_bench_rocprofv3_kernel_stats.csv
  rm -rf $O/prof
  cd $R
  timeout 900 bash tools/pmc_passes.sh c2 > $O/pmc_c2.log 2>&1
  cp $R/gpurun_out/pmc_c2/pmc_c2.json $O/${RD}_pmc_c2.json
  timeout 600 python bench.py > $O/${RD}_default_bench.json.txt 2> $O/bench_default.err; echo "bench rc=$?"
  python tools/bench_summary.py $O/${RD}_default_bench.json.txt | head -4
  head -8 $O/${RD}_c2_only_bench_rocprofv3_kernel_stats.csv | cut -c1-150
  exit 0
fi
if [ "$1" != "noprof_tests" ]; then
timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > $O/${RD}_gpu_tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/${RD}_gpu_tests.log
tail -5 $O/${RD}_gpu_tests.log
fi
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/${RD}_smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/${RD}_smoke.log
timeout 600 python bench.py > $O/${RD}_default_bench.json.txt 2> $O/bench_default.err; echo "bench rc=$?"
tail -c 600 $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/c2 -o trace -- python $R/bench.py --no-batch32 --no-cpu-baseline --no-host-api --no-extras --steps 50 > $O/${RD}_c2_only_bench_under_rocprof.json.txt 2> $O/prof_c2.err
cp $(find $O/prof/c2 -name '*kernel_stats.csv' | head -1) $O/${RD}_c2_only_bench_rocprofv3_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/c3 -o trace -- python $R/bench.py --workload c3 --no-cpu-baseline --no-host-api > $O/${RD}_c3_only_bench_under_rocprof.json.txt 2> $O/prof_c3.err
cp $(find $O/prof/c3 -name '*kernel_stats.csv' | head -1) $O/${RD}_c3_only_bench_rocprofv3_kernel_stats.csv
rm -rf $O/prof
cd $R
for w in c3 c4 c5 m2 m3; do
  timeout 300 python bench.py --workload $w --no-cpu-baseline --no-host-api > $O/${RD}_${w}_bench.json.txt 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
# the self-launching multi-rank entry point: two replicas on this one device, host-side (gloo) barrier, batch256_sharded over 2 ranks
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/${RD}_gpus2_selflaunch_bench.json.txt 2> $O/bench_gpus2.err; echo "bench --gpus 2 (self-launch) rc=$?"
for w in c3 c5; do
  timeout 300 python bench.py --workload $w --precision bf16x3 --no-cpu-baseline --no-host-api > $O/${RD}_${w}_bf16x3_bench.json.txt 2> $O/bench_${w}b.err; echo "bench $w bf16x3 rc=$?"
done
cd $R
for w in c2 c3; do
  timeout 900 bash tools/pmc_passes.sh $w > $O/pmc_$w.log 2>&1
  cp $R/gpurun_out/pmc_$w/pmc_$w.json $O/${RD}_pmc_$w.json
  rm -rf $R/gpurun_out/pmc_$w
done
bf16x3_profiles
# step timelines of the three single-stage persistent programs (tracing build)
for p in enc dp flow; do PS_DETAIL=1 timeout 200 python tools/ps_trace.py $p > $O/${RD}_ps_trace_$p.txt 2>&1; tail -1 $O/${RD}_ps_trace_$p.txt; done
ls -la $O
python tools/bench_summary.py $O/${RD}_default_bench.json.txt
