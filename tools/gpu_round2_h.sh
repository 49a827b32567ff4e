#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
(timeout 1200 python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/pytest_h.log 2>&1; echo "exit $?" >> gpurun_out/pytest_h.log)
for v in 0 2 4; do
(SJHIP_S1_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_stage1.py tests/test_gpu_parse.py -m gpu -x -q -k "not full_size and not stream" > gpurun_out/pytest_h_v$v.log 2>&1; echo "exit $?" >> gpurun_out/pytest_h_v$v.log)
done
(VARIANTS="1 4" timeout 600 python tools/s1_experiment.py > gpurun_out/s1_tail.log 2>&1; echo "exit $?" >> gpurun_out/s1_tail.log)
(SJHIP_S1_NO_TAIL=1 VARIANTS="1" COPIES="426" timeout 600 python tools/s1_experiment.py > gpurun_out/s1_notail.log 2>&1; echo "exit $?" >> gpurun_out/s1_notail.log)
(timeout 300 python tools/parse_latency.py > gpurun_out/latency_tail.log 2>&1; SJHIP_S1_NO_TAIL=1 timeout 300 python tools/parse_latency.py > gpurun_out/latency_notail.log 2>&1)
for f in gpurun_out/pytest_h.log gpurun_out/pytest_h_v0.log gpurun_out/pytest_h_v2.log gpurun_out/pytest_h_v4.log; do echo "== $f"; tail -n 5 $f | cut -c1-300; done
for f in gpurun_out/s1_tail.log gpurun_out/s1_notail.log; do echo "== $f"; grep '^{' $f | cut -c1-330; done
for f in gpurun_out/latency_tail.log gpurun_out/latency_notail.log; do echo "== $f"; tail -n 8 $f | cut -c1-300; done
