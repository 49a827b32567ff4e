#!/bin/bash
# MarshalJSON of the two bench workloads: key flags from the parser or recovered, tile-kernel variants (SJHIP_MS_VARIANT)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-marshal_var}
mkdir -p $OUT
cd $REPO
for v in 0 1 3 6 7; do
  for w in twitter parking; do
    echo "variant $v $w" | tee -a $OUT/times.txt
    SJHIP_MS_VARIANT=$v timeout 120 python tools/marshal_loop.py $w 5 kf 2>&1 | grep -E "marshal_json" | tee -a $OUT/times.txt
  done
done
echo "no key flags" | tee -a $OUT/times.txt
timeout 120 python tools/marshal_loop.py parking 5 2>&1 | grep -E "marshal_json|serialize" | tee -a $OUT/times.txt
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/trace_parking -o p -- python $REPO/tools/marshal_loop.py parking 3 kf > /dev/null 2>&1
python $REPO/tools/kernel_times.py $OUT/trace_parking/p_results.db "" | grep -v "k_s1\|stage1" | tee $OUT/kernels.txt
