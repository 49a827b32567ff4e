#!/bin/bash
# round-2 GPU batch A: stage-1 variants + timelines, the barrier-free kernel under the parity tests, the whole GPU suite
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
nproc > gpurun_out/nproc.txt
(timeout 900 python tools/s1_experiment.py > gpurun_out/s1_experiment.log 2>&1; echo "exit $?" >> gpurun_out/s1_experiment.log)
(SJHIP_S1_VARIANT=3 timeout 900 python -m pytest tests/test_gpu_stage1.py tests/test_gpu_parse.py tests/test_gpu_quirks.py -m gpu -x -q -k "not full_size" > gpurun_out/pytest_v3.log 2>&1; echo "exit $?" >> gpurun_out/pytest_v3.log)
(timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu_r2a.log 2>&1; echo "exit $?" >> gpurun_out/pytest_gpu_r2a.log)
tail -5 gpurun_out/s1_experiment.log gpurun_out/pytest_v3.log gpurun_out/pytest_gpu_r2a.log
