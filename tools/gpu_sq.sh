#!/bin/bash
# SQ counters of the parse kernels on one workload: tools/gpu_sq.sh <name> <twitter|parking>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-sq}
W=${2:-parking}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq1 -o p -- python $REPO/tools/parse_loop.py $W 3 > $OUT/sq1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/sq2 -o p -- python $REPO/tools/parse_loop.py $W 3 > $OUT/sq2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_IFETCH SQ_ACTIVE_INST_EXP_GDS -d $OUT/sq3 -o p -- python $REPO/tools/parse_loop.py $W 3 > $OUT/sq3.log 2>&1
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "pmc" $OUT/summary.txt | sed "s/(sj::S2Dev[^)]*)//" | grep -E "k_s2_emit|k_measure|stage1" | cut -c1-170
