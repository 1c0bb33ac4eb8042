#!/bin/bash
# rocprofv3 PMC passes for one bench workload (separate runs per counter group: FETCH_SIZE and WRITE_SIZE do
# not fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"; gpurun refuses --pmc combined with the
# sys/hip/hsa trace domains, so only kernel-dispatch counter data is collected).
#   tools/pmc_passes.sh c3                          -> gpurun_out/pmc_c3/*.csv ; summarize with profiles/summarize_pmc.py
#   tools/pmc_passes.sh c3 "--precision bf16x3" _bf16x3  -> gpurun_out/pmc_c3_bf16x3/ (extra bench arguments, output suffix)
set -e
W=${1:-c3}
EXTRA=${2:-}
SUF=${3:-}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_$W$SUF
mkdir -p $OUT
CMD="python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline --no-batch32 --no-host-api --no-extras --min-seconds 0 --no-graph $EXTRA"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq1 -- $CMD > /dev/null 2>&1 || echo pass1 failed
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $OUT -o sq2 -- $CMD > /dev/null 2>&1 || echo pass2 failed
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $CMD > /dev/null 2>&1 || echo pass3 failed
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $CMD > /dev/null 2>&1 || echo pass4 failed
# calibration of FETCH_SIZE / WRITE_SIZE for 4 B/lane and 16 B/lane accesses (1 GiB streams)
if [ -x $R/tools/pmc_calib ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o calibfetch -- $R/tools/pmc_calib > /dev/null 2>&1 || echo calib pass1 failed
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o calibwrite -- $R/tools/pmc_calib > /dev/null 2>&1 || echo calib pass2 failed
fi
python $R/profiles/summarize_pmc.py $OUT $W$SUF > $OUT/pmc_$W$SUF.json
ls $OUT
