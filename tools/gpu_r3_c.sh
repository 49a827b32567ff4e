#!/bin/bash
# round 3: GPU test-suite, whole-parse wall times, kernel traces of the two BASELINE workloads -> gpurun_out/$1
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r3c}
mkdir -p $OUT
cd $REPO
timeout 200 python tools/parse_time.py 2>&1 | tee $OUT/parse_time.txt
timeout 300 python -m pytest tests/test_gpu_parse.py tests/test_gpu_quirks.py tests/test_gpu_fuzz.py -m gpu -q --maxfail=8 2>&1 | tail -30 > $OUT/tests.txt
tail -12 $OUT/tests.txt
cd /tmp && export TMPDIR=/tmp
for w in twitter parking; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o p -- python $REPO/tools/parse_loop.py $w 6 > $OUT/trace_$w.log 2>&1
  python $REPO/tools/kernel_times.py $OUT/trace_$w/p_results.db | tee $OUT/kernels_$w.txt
done
