#!/bin/bash
# stage-1 kernel time of every variant at 64 MiB / 256 MiB / 1 GiB (tools/s1_time.py)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for c in 13 107 426 1700; do
  for v in 1; do
    echo -n "copies=$c variant=$v  "; COPIES=$c SJHIP_S1_VARIANT=$v python tools/s1_time.py 2>&1 | grep -v amdgpu.ids
  done
done
