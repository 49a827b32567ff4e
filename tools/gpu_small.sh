#!/bin/bash
# kernel timeline of Parse() of one small document, device-resident and host -> host, per fixture -> gpurun_out/$1/timeline_*.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-small}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in twitter twitterescaped canada; do
  for tool in small_dev_loop small_doc_trace; do
    python $REPO/tools/$tool.py $f 200 2>&1 | grep -v amdgpu.ids
    rm -rf $OUT/t
    timeout 120 rocprofv3 --kernel-trace -d $OUT/t -o p -- python $REPO/tools/$tool.py $f 10 > $OUT/log 2>&1
    python $REPO/tools/timeline.py $OUT/t/p_results.db 1 | tee $OUT/timeline_${f}_$tool.txt
  done
done
