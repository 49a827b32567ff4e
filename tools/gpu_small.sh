#!/bin/bash
# kernel timeline of Parse() of one small document (host -> host), per fixture
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-small}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in twitter twitterescaped canada; do
  python $REPO/tools/small_doc_trace.py $f 200 2>&1 | grep -v amdgpu.ids
  rm -rf $OUT/t_$f
  timeout 120 rocprofv3 --kernel-trace -d $OUT/t_$f -o p -- python $REPO/tools/small_doc_trace.py $f 10 > $OUT/log 2>&1
  python $REPO/tools/timeline.py $OUT/t_$f/p_results.db 1 | tee $OUT/timeline_$f.txt
done
