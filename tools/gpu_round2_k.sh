#!/bin/bash
# A/B of compile-time stage-1 variants built into build_ab/ (SJHIP_LIB selects the library)
mkdir -p gpurun_out
for lib in $(ls build_ab/*.so) $(ls build_ab/*.so); do
  echo "== lib ${lib:-default}"
  for c in ${COPIES_LIST:-426 1700}; do
    SJHIP_LIB=${lib:+$PWD/$lib} COPIES=$c timeout 120 python tools/s1_time.py 2>&1 | tail -1
  done
done
for lib in $SKIPTEST; do
  SJHIP_LIB=$PWD/$lib timeout 300 python -m pytest tests/test_gpu_stage1.py -x -q -m gpu 2>&1 | tail -2
done
