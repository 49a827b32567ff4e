#!/bin/bash
# parts of k_ms_tile left out one at a time (SJ_EXP build, SJHIP_MS_EXP): MarshalJSON of configs[4]'s and configs[1]'s tape, wall ms
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export SJHIP_LIB=$PWD/build_ab/libsjhip_exp.so
{
for w in parking twitter; do
for bits in 0 1 2 4 8 16 32 3 19 0; do
  echo -n "$w exp=$bits  "
  SJHIP_MS_EXP=$bits timeout 200 python tools/marshal_loop.py $w 5 kf 2>&1 | grep marshal_json
done
done
} 2>&1 | tee gpurun_out/exp_marshal.txt
