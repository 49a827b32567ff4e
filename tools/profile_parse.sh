#!/bin/bash
# rocprofv3 kernel trace + stats of the whole-parse legs (stage 1 + stage 2 kernels).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-prof_parse}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace -o parse -- python $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
