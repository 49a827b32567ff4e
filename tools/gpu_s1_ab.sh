#!/bin/bash
# same-box A/B of the plain stage-1 kernel of several builds: tools/gpu_s1_ab.sh <a.so> <b.so> ...   (hipEvent kernel time, ms, on the
# 64 MiB / 256 MiB / 1 GiB documents: tools/s1_time.py; two alternating rounds)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for r in 1 2; do
  for lib in "$@"; do
    echo -n "$lib  "
    for c in 107 426 1700; do
      COPIES=$c SJHIP_LIB=$PWD/$lib timeout 200 python tools/s1_time.py 2>&1 | grep -v amdgpu.ids | awk -v c=$c '{printf "x%s %s ms; ", c, $1}'
    done
    echo
  done
done
