#!/bin/bash
# same-box A/B of builds of the library, both copy modes: tools/gpu_ab_modes.sh <rounds> <a.so> <b.so> ...
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
R=$1; shift
for r in $(seq 1 $R); do
  for lib in "$@"; do
    echo "== $lib (round $r)"
    SJHIP_LIB=$REPO/$lib timeout 300 python tools/nocopy_time.py 2>&1 | grep -v amdgpu.ids | head -4 | tr '\n' ';'; echo
  done
done
