#!/bin/bash
# A/B of the stage-1 tile shapes on the GPU box: parity tests + kernel time per variant.
# usage: tools/gpu_s1_ab.sh "0 1 2 3 4"
mkdir -p gpurun_out
for v in ${1:-0}; do
  echo "== variant $v"
  SJHIP_S1_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_stage1.py -x -q -m gpu 2>&1 | tail -3
  SJHIP_S1_VARIANT=$v timeout 300 python bench.py --steps 20 --warmup 3 --stage1-only --no-cpu-baseline 2>gpurun_out/ab_$v.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['input_GBps'])"
done
