#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
(VARIANTS="1 4" timeout 600 python tools/s1_experiment.py > gpurun_out/s1_experiment.log 2>&1; echo "exit $?" >> gpurun_out/s1_experiment.log)
(timeout 600 python tools/stream_bench.py > gpurun_out/stream_bench.log 2>&1; echo "exit $?" >> gpurun_out/stream_bench.log)
for f in gpurun_out/s1_experiment.log gpurun_out/stream_bench.log; do echo "== $f"; grep -v '"stream"' $f | tail -n 16 | cut -c1-420; done
