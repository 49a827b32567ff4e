#!/bin/bash
# round 5, first GPU call: the GPU suite on the plane kernels, then A/B of the token-pass variants and kernel traces
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r5a
mkdir -p $OUT
cd $REPO
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/pytest.log
for v in 1 0; do
  echo "== SJHIP_S2_VARIANT=$v"
  SJHIP_S2_VARIANT=$v timeout 300 python tools/parse_time.py 2>&1 | tee $OUT/parse_time_v$v.log
done
bash tools/gpu_trace_parse.sh r5a/tw twitter
bash tools/gpu_trace_parse.sh r5a/pk parking
