#!/bin/bash
# A/B of SJHIP_S2_OVERLAP (parse_api.hip): wall time of the whole parse per mode
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-ab_overlap}
mkdir -p $OUT
cd $REPO
for m in 1 3 2 1 3; do
  echo "== SJHIP_S2_OVERLAP=$m" | tee -a $OUT/ab.txt
  SJHIP_S2_OVERLAP=$m timeout 200 python tools/parse_time.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/ab.txt
done
