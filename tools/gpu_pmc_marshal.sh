#!/bin/bash
# where the cycles of MarshalJSON's tile kernel go: SQ counters of marshal_loop.py (parking, key flags from the parser)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-pmc_marshal}
W=${2:-parking}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_FLAT" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $set -d $OUT/set$i -o p -- python $REPO/tools/marshal_loop.py $W 2 kf > $OUT/set$i.log 2>&1 || tail -3 $OUT/set$i.log
done
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null 2>&1
grep -E "pmc " $OUT/summary.txt | grep -E "k_ms_|k_ser|k_des" | cut -c1-260
