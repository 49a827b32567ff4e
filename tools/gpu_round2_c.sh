#!/bin/bash
# round-2 GPU batch C: query tests, stream bench, whole-parse profile (trace + PMC), bench line, full GPU suite
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
(timeout 600 python -m pytest tests/test_gpu_query.py -m gpu -x -q > gpurun_out/pytest_query.log 2>&1; echo "exit $?" >> gpurun_out/pytest_query.log)
(timeout 600 python tools/stream_bench.py > gpurun_out/stream_bench.log 2>&1; echo "exit $?" >> gpurun_out/stream_bench.log)
(timeout 900 bash tools/profile_parse_r2.sh prof_parse_r2 > gpurun_out/prof_parse_r2.log 2>&1; echo "exit $?" >> gpurun_out/prof_parse_r2.log)
(timeout 900 python bench.py > gpurun_out/bench_r2.log 2>&1; echo "exit $?" >> gpurun_out/bench_r2.log)
(timeout 1500 python -m pytest tests -m gpu -q --durations=5 > gpurun_out/pytest_gpu_r2c.log 2>&1; echo "exit $?" >> gpurun_out/pytest_gpu_r2c.log)
for f in gpurun_out/pytest_query.log gpurun_out/stream_bench.log gpurun_out/prof_parse_r2.log gpurun_out/bench_r2.log gpurun_out/pytest_gpu_r2c.log; do echo "== $f"; tail -n 12 $f | cut -c1-2500; done
