#!/bin/bash
# PMC passes for the dominant kernel at workload c3 (separate runs per counter group; gpurun refuses
# --pmc together with the sys/hip/hsa trace domains, so only kernel dispatch data is collected).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_c3
mkdir -p $OUT
CMD="python $R/bench.py --workload ${1:-c3} --steps 2 --warmup 1 --no-cpu-baseline --no-batch32"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT -o sq1 -- $CMD > /dev/null 2>&1 || echo pass1 failed
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT -o sq2 -- $CMD > /dev/null 2>&1 || echo pass2 failed
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- $CMD > /dev/null 2>&1 || echo pass3 failed
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- $CMD > /dev/null 2>&1 || echo pass4 failed
ls -la $OUT
