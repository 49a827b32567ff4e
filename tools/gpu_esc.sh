#!/bin/bash
# SQ counters of the string kernels on one small escape-heavy document
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-esc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS -d $OUT/pmc -o p -- python $REPO/tools/small_doc_trace.py twitterescaped 6 > $OUT/log 2>&1
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "pmc|kernel " $OUT/summary.txt | sed "s/(sj::S2Dev[^)]*)//" | grep -E "k_measure|k_str_emit" | sed "s/ *dispatches=[0-9]*//; s/min=.*//" | cut -c1-150
