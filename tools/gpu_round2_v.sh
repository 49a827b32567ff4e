#!/bin/bash
# final validation of the round: full GPU suite, variants, smoke, bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/v_tests.txt; cat gpurun_out/v_tests.txt
for v in 0 3 4; do SJHIP_S1_VARIANT=$v timeout 300 python -m pytest tests/test_gpu_stage1.py -x -q -m gpu 2>&1 | tail -1; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/v_bench.json 2> gpurun_out/v_bench.err; tail -c 300 gpurun_out/v_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/v_bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, {k:d["roofline"][k] for k in ("kernel_ms","frac","read_frac","input_frac")}, d["roofline"]["at_1GiB"]["kernel_ms"], d["roofline"]["at_1GiB"]["frac"])
print("full", d["full_parse"]["ms"], d["full_parse"]["roofline"]["frac"], "nd", d["ndjson"]["ms"], d["ndjson"]["roofline"]["frac"])
print("query", d["query"]["count_ms"], d["query"]["filter_ms"], "ser", d["serialize"]["ms"], "ms", d["marshal_json"]["ms"], "stream", d["stream"]["GBps"])
PY
