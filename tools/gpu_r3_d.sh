#!/bin/bash
# string-path check: GPU string/parse tests, small-document timelines, kernel times of the two big workloads
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r3d}
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parse.py -m gpu -q --maxfail=8 2>&1 | tail -8 | tee $OUT/tests.txt
timeout 200 python tools/parse_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/parse_time.txt
cd /tmp && export TMPDIR=/tmp
for f in twitter twitterescaped; do
  rm -rf $OUT/t_$f
  timeout 120 rocprofv3 --kernel-trace -d $OUT/t_$f -o p -- python $REPO/tools/small_doc_trace.py $f 10 > $OUT/log 2>&1
  python $REPO/tools/timeline.py $OUT/t_$f/p_results.db 1 | tee $OUT/timeline_$f.txt
done
for w in twitter parking; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o p -- python $REPO/tools/parse_loop.py $w 6 > $OUT/trace_$w.log 2>&1
  python $REPO/tools/kernel_times.py $OUT/trace_$w/p_results.db | tee $OUT/kernels_$w.txt
done
