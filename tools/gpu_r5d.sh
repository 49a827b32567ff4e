#!/bin/bash
# round 5: the masks-only stage 1 / soff[] form of the string path: parity subset, wall times, kernel traces
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5d
mkdir -p $OUT
cd $REPO
timeout 600 python -m pytest tests/test_gpu_parse.py tests/test_gpu_quirks.py tests/test_gpu_fuzz.py -m gpu -x -q > $OUT/pytest_subset.log 2>&1
echo "pytest subset rc=$?"; tail -4 $OUT/pytest_subset.log
for it in 8 16; do
  echo "== SJHIP_S2_ITEMS=$it"
  SJHIP_S2_ITEMS=$it timeout 300 python tools/parse_time.py 2>&1 | grep -v amdgpu.ids | head -6
done
bash tools/gpu_trace_parse.sh r5d/tw twitter | head -12
bash tools/gpu_trace_parse.sh r5d/pk parking | head -12
rm -f $OUT/*/trace/*.db
