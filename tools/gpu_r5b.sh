#!/bin/bash
# round 5: A/B of the emit variants (parse wall time + kernel trace), then the GPU suite
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5b
mkdir -p $OUT
cd $REPO
for cfg in "1 8" "1 16" "0 8"; do
  set -- $cfg
  echo "== SJHIP_S2_VARIANT=$1 SJHIP_S2_ITEMS=$2"
  SJHIP_S2_VARIANT=$1 SJHIP_S2_ITEMS=$2 timeout 300 python tools/parse_time.py 2>&1 | grep -v amdgpu.ids | head -2
done
bash tools/gpu_trace_parse.sh r5b/tw twitter | head -8
bash tools/gpu_trace_parse.sh r5b/pk parking | head -8
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
