#!/bin/bash
# MarshalJSON: the marshal tests on the tree's library, then same-box A/B of MarshalJSON wall time for $LIBS (alternating)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{
timeout 1200 python -m pytest tests/test_gpu_marshal.py -m gpu -x -q 2>&1 | tail -6
for r in 1 2 3; do
  for lib in $LIBS; do
    for w in parking twitter; do
      echo -n "$lib $w  "
      SJHIP_LIB=$PWD/$lib timeout 200 python tools/marshal_loop.py $w 5 kf 2>&1 | grep marshal_json
    done
  done
done
} 2>&1 | tee gpurun_out/${OUTNAME:-ab_marshal_libs}.txt
