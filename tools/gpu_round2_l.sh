#!/bin/bash
mkdir -p gpurun_out
for c in 426 1700; do COPIES=$c timeout 120 python tools/s1_time.py 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_stage1.py tests/test_gpu_parse.py -x -q -m gpu 2>&1 | tail -3
for v in 0 2 3 4; do SJHIP_S1_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_stage1.py -x -q -m gpu 2>&1 | tail -1; SJHIP_S1_VARIANT=$v timeout 100 python tools/s1_time.py | tail -1; done
