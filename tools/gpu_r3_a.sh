#!/bin/bash
# round 3, call A: GPU test-suite, whole-parse wall times and kernel traces of the two BASELINE workloads
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r3a
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests -m gpu -q -x --maxfail=8 2>&1 | tail -40 > $OUT/tests.txt
cat $OUT/tests.txt | tail -15
timeout 200 python tools/parse_time.py > $OUT/parse_time.txt 2>&1
cat $OUT/parse_time.txt
cd /tmp && export TMPDIR=/tmp
for w in twitter parking; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o p -- python $REPO/tools/parse_loop.py $w 6 > $OUT/trace_$w.log 2>&1
  tail -2 $OUT/trace_$w.log
  f=$(ls $OUT/trace_$w/*/p_kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cut -d, -f1-6 $f | head -16 | tee $OUT/stats_$w.txt
done
