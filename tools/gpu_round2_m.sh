#!/bin/bash
mkdir -p gpurun_out
for v in ${VARS:-1 5 6 0}; do
  echo "== variant $v"
  SJHIP_S1_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_stage1.py -x -q -m gpu 2>&1 | tail -1
  for c in 426 1700; do SJHIP_S1_VARIANT=$v COPIES=$c timeout 120 python tools/s1_time.py 2>&1 | tail -1; done
done
