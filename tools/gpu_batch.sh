#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-batch}
mkdir -p $OUT
cd $REPO
true
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d.get('extra_error')); print(json.dumps(d.get('batch'),indent=1)); print(json.dumps({k:(v['host_to_host_us'],v['device_us']) for k,v in d['single_documents'].items()}))
print(d['value'], d['value_stage1_stage2'], d['full_parse']['ms'], d['ndjson'].get('ms'))
" | tee $OUT/bench.txt
