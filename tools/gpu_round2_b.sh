#!/bin/bash
# round-2 GPU batch: query + stream tests and benches, whole GPU suite
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
(timeout 600 python -m pytest tests/test_gpu_query.py tests/test_gpu_parse.py -m gpu -x -q -k "query or stream or hond or record or full_size_count" > gpurun_out/pytest_query.log 2>&1; echo "exit $?" >> gpurun_out/pytest_query.log)
(timeout 600 python tools/stream_bench.py > gpurun_out/stream_bench.log 2>&1; echo "exit $?" >> gpurun_out/stream_bench.log)
(timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu_r2c.log 2>&1; echo "exit $?" >> gpurun_out/pytest_gpu_r2c.log)
for f in gpurun_out/pytest_query.log gpurun_out/stream_bench.log gpurun_out/pytest_gpu_r2c.log; do echo "== $f"; tail -n 14 $f | cut -c1-700; done
