#!/bin/bash
# nb kernel rework (concurrent duties, chained tickets, eager flatten): parity for variants 3/4, then A/B + traces
mkdir -p gpurun_out
for v in 4 3; do
  echo "== variant $v"
  SJHIP_S1_VARIANT=$v timeout 400 python -m pytest tests/test_gpu_stage1.py tests/test_gpu_parse.py -x -q -m gpu 2>&1 | tail -3
done
VARIANTS="1 3 4" timeout 400 python tools/s1_experiment.py > gpurun_out/s1_experiment_i.log 2>&1
tail -30 gpurun_out/s1_experiment_i.log
