#!/bin/bash
# A/B of the stage-1 block shapes on the GPU box: parity tests + kernel time per variant.
# usage: tools/gpu_s1_ab.sh "0 1"   (0: 512x2, 1: 1024x2 = default)
mkdir -p gpurun_out
for v in ${1:-0}; do
  echo "== variant $v"
  SJHIP_S1_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_stage1.py -x -q -m gpu 2>&1 | tail -3
  SJHIP_S1_VARIANT=$v timeout 120 python tools/s1_time.py
done
