#!/bin/bash
# round-2 GPU batch B: stage-1 variants with the claimed serial duty + timelines, parity tests under variants 3 / 4
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
(timeout 900 python tools/s1_experiment.py > gpurun_out/s1_experiment.log 2>&1; echo "exit $?" >> gpurun_out/s1_experiment.log)
for v in 6; do
(SJHIP_S1_VARIANT=$v timeout 900 python -m pytest tests/test_gpu_stage1.py tests/test_gpu_parse.py tests/test_gpu_quirks.py -m gpu -x -q > gpurun_out/pytest_v$v.log 2>&1; echo "exit $?" >> gpurun_out/pytest_v$v.log)
done
(timeout 600 python tools/stream_bench.py > gpurun_out/stream_bench.log 2>&1; echo "exit $?" >> gpurun_out/stream_bench.log)
for f in gpurun_out/stream_bench.log gpurun_out/s1_experiment.log gpurun_out/pytest_v6.log; do echo "== $f"; tail -n 6 $f | cut -c1-600; done
