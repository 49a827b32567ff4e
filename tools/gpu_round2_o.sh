#!/bin/bash
# kernel trace of the whole parse (twitter only, quick) -> gpurun_out/o_trace/summary.txt
REPO=$(pwd)
OUT=$REPO/gpurun_out/o_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in ${W:-twitter}; do
  timeout 150 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o p -- python $REPO/tools/parse_loop.py $w 6 > $OUT/trace_$w.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "kernel " $OUT/summary.txt | sed 's/void sj:://; s/(anonymous namespace):://' | cut -c1-150 | head -40
