#!/bin/bash
# same-box A/B of builds ($LIBS): plain stage 1 at three sizes, the whole parse in both copy modes
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{
bash tools/gpu_s1_ab.sh $LIBS
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in $LIBS; do
    echo -n "$lib  "
    SJHIP_LIB=$PWD/$lib timeout 300 python tools/nocopy_time.py 2>&1 | grep -v amdgpu.ids | awk '{printf "%s/%s %s; ", $1, $2, $3}'
    echo
  done
done
} 2>&1 | tee gpurun_out/${OUTNAME:-ab_libs_s1}.txt
