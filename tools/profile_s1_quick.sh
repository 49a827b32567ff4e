#!/bin/bash
# Quick PMC passes over the stage-1 kernel (tools/s1_time.py as the workload).
# Usage (GPU box, repo root): tools/profile_s1_quick.sh <outdir-under-gpurun_out> [variant]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-profq}
export SJHIP_S1_VARIANT=${2:-0}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/tools/s1_time.py"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o s1 -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/pmc_sq -o s1 -- $CMD > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o s1 -- $CMD > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o s1 -- $CMD > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o s1 -- $CMD > $OUT/pmc_write.log 2>&1
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "stage1_kernel" $OUT/summary.txt
