#!/bin/bash
# same-box A/B of two builds of the library: tools/gpu_ab.sh <a.so> <b.so> [rounds]   (wall time of the whole parse, both workloads;
# boxes differ by several percent between calls, so variants are only compared inside one call, alternating)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
A=$1; B=$2; R=${3:-2}
for r in $(seq 1 $R); do
  for lib in $A $B; do
    echo "== $lib (round $r)"
    SJHIP_LIB=$REPO/$lib timeout 300 python tools/parse_time.py 2>&1 | grep -v amdgpu.ids | head -2 | awk '{printf "%s %s ms; ", $1, $(NF-3)}'; echo
    SJHIP_LIB=$REPO/$lib timeout 300 python tools/small_time.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
