#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stage1.py tests/test_gpu_parse.py tests/test_gpu_quirks.py -x -q -m gpu 2>&1 | tail -3
for v in 3 4; do SJHIP_S1_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_stage1.py -x -q -m gpu 2>&1 | tail -1; done
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --stage1-only 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['kernel_ms'], d['roofline']['frac'])"
