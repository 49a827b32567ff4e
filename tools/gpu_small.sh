#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-small}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for f in twitter canada; do
for sb in default 0; do
  if [ $sb = default ]; then unset SJHIP_SMALL_BYTES; else export SJHIP_SMALL_BYTES=$sb; fi
  python $REPO/tools/small_doc_trace.py $f 200
  rm -rf $OUT/t_${f}_$sb
  timeout 120 rocprofv3 --kernel-trace -d $OUT/t_${f}_$sb -o p -- python $REPO/tools/small_doc_trace.py $f 10 > $OUT/log 2>&1
  python $REPO/tools/timeline.py $OUT/t_${f}_$sb/p_results.db 1
done
done
