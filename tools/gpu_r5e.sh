#!/bin/bash
# round 5: WithCopyStrings(false) on the masks (byte-parallel selective copy): parity subset, wall times of both modes, traces
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5e
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests/test_gpu_parse.py tests/test_gpu_quirks.py tests/test_gpu_fuzz.py tests/test_gpu_multi.py tests/test_gpu_batch.py -m gpu -x -q > $OUT/pytest_subset.log 2>&1
echo "pytest subset rc=$?"; tail -6 $OUT/pytest_subset.log
timeout 300 python tools/nocopy_time.py 2>&1 | grep -v amdgpu.ids | head -12
bash tools/gpu_trace_parse.sh r5e/tw_nocopy twitter nocopy | head -14
bash tools/gpu_trace_parse.sh r5e/pk_nocopy parking nocopy | head -14
rm -f $OUT/*/trace/*.db
