#!/bin/bash
# string general path: parse tests, small-document timelines (twitter, twitterescaped), big-document kernel times
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-r3e}
mkdir -p $OUT
cd $REPO
timeout 300 python -m pytest tests/test_gpu_parse.py tests/test_gpu_quirks.py tests/test_gpu_fuzz.py -m gpu -q --maxfail=8 2>&1 | tail -8 | tee $OUT/tests.txt
timeout 200 python tools/parse_time.py 2>&1 | grep -v amdgpu.ids | tee $OUT/parse_time.txt
cd /tmp && export TMPDIR=/tmp
for f in twitter twitterescaped; do
  rm -rf $OUT/t_$f
  timeout 120 rocprofv3 --kernel-trace -d $OUT/t_$f -o p -- python $REPO/tools/small_doc_trace.py $f 10 > $OUT/log 2>&1
  python $REPO/tools/timeline.py $OUT/t_$f/p_results.db 1 | grep -E "k_measure|k_str_emit|total" | tee -a $OUT/timeline.txt
done
