#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
(timeout 900 python -m pytest tests/test_gpu_parse.py tests/test_gpu_quirks.py tests/test_gpu_query.py -m gpu -x -q > gpurun_out/pytest_e.log 2>&1; echo "exit $?" >> gpurun_out/pytest_e.log)
(timeout 600 bash tools/profile_parse_r2.sh prof_parse_r2b > gpurun_out/prof_parse_r2b.log 2>&1; echo "exit $?" >> gpurun_out/prof_parse_r2b.log)
for f in gpurun_out/pytest_e.log; do echo "== $f"; tail -n 8 $f | cut -c1-400; done
python - <<'PY'
import json
d=json.load(open('gpurun_out/prof_parse_r2b/r02_parse_kernels.json'))
for w in ('twitter_x426','parking_x1000_nd'):
    print(w, d[w]['sum_us'], {k:v for k,v in list(d[w]['us_per_parse'].items())[:9]})
PY
