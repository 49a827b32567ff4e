#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-marshal}
mkdir -p $OUT
cd $REPO
if [ "${2:-}" = "test" ]; then timeout 200 python -m pytest tests/test_gpu_marshal.py tests/test_gpu_serialize.py -m gpu -q --maxfail=5 2>&1 | tail -8 | tee $OUT/tests.txt; fi
cd /tmp && export TMPDIR=/tmp
for w in twitter parking; do
  timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o p -- python $REPO/tools/marshal_loop.py $w 3 2>&1 | grep -E "marshal_json|serialize" | tee -a $OUT/times.txt
  python $REPO/tools/kernel_times.py $OUT/trace_$w/p_results.db k_ | grep -E "k_ms|k_tw|k_ser" | tee -a $OUT/kernels.txt
done
