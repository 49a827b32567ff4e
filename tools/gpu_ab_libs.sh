#!/bin/bash
# parse tests on the tree's library, then same-box A/B of the whole parse in both copy modes for $LIBS (alternating)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests/test_gpu_parse.py tests/test_gpu_quirks.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -3
for r in $(seq 1 ${ROUNDS:-3}); do
  for lib in $LIBS; do
    echo -n "$lib  "
    SJHIP_LIB=$PWD/$lib timeout 300 python tools/nocopy_time.py 2>&1 | grep -v amdgpu.ids | awk '{printf "%s/%s %s; ", $1, $2, $3}'
    echo
  done
done
} 2>&1 | tee gpurun_out/${OUTNAME:-ab_libs}.txt
