#!/bin/bash
# kernel trace of the whole parse: tools/gpu_trace_parse.sh <name> <twitter|parking> [nocopy]   -> gpurun_out/<name>/
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/trace -o p -- python $REPO/tools/parse_loop.py $2 6 ${3:-copy} > $OUT/trace.log 2>&1
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "^==|kernel " $OUT/summary.txt | sed "s/(sj::S2Dev[^)]*)//" | cut -c1-160
