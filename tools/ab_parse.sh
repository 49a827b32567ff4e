#!/bin/bash
# A/B of library builds (simdjson-go_amd/exp_<name>.so): per-kernel times of the whole-parse legs under rocprofv3.
# usage: tools/ab_parse.sh name...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  echo "== $n"
  OUT=$REPO/gpurun_out/exp_$n
  rm -rf $OUT; mkdir -p $OUT
  SJHIP_LIB=$REPO/simdjson-go_amd/exp_$n.so rocprofv3 --kernel-trace -d $OUT -o t -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/log 2>&1
  python $REPO/tools/kernel_times.py $OUT/t_results.db "sj::k_" | cut -c1-120
done
