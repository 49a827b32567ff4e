#!/bin/bash
# HBM counters (+ VALU instructions) of every kernel of one parse: tools/gpu_pmc_parse.sh <name> <twitter|parking> [nocopy]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c -d $OUT/$c -o p -- python $REPO/tools/parse_loop.py $2 3 ${3:-copy} > $OUT/$c.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "pmc " $OUT/summary.txt | sed "s/(sj::S2Dev[^)]*)//" | grep -v "copyBuffer\|prepare\|min_upper" | cut -c1-150
