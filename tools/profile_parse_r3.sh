#!/bin/bash
# Whole-parse profile of the two BASELINE workloads: kernel trace + stats, HBM counters (FETCH_SIZE / WRITE_SIZE in
# separate PMC passes, MI355X_MICROARCH.md) and two SQ passes.  Usage (GPU box): tools/profile_parse_r3.sh <name>
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-prof_parse_r3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in twitter parking; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o p -- python $REPO/tools/parse_loop.py $w 6 > $OUT/trace_$w.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$w -o p -- python $REPO/tools/parse_loop.py $w 3 > $OUT/fetch_$w.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$w -o p -- python $REPO/tools/parse_loop.py $w 3 > $OUT/write_$w.log 2>&1
done
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/sq1_twitter -o p -- python $REPO/tools/parse_loop.py twitter 3 > $OUT/sq1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $OUT/sq2_twitter -o p -- python $REPO/tools/parse_loop.py twitter 3 > $OUT/sq2.log 2>&1
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "pmc|kernel " $OUT/summary.txt | sed "s/(sj::S2Dev[^)]*)//" | grep -E "k_str_emit|k_s2_emit|k_str_masks|k_br_match|k_numbers|stage1" | grep -v fillBuffer | head -150
