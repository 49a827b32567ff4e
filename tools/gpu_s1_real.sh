#!/bin/bash
# kernel durations of real sjhip_stage1_device calls (rocprofv3) for the libraries named on the command line: tools/gpu_s1_real.sh <lib.so>...
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  for c in 13 426; do
    rm -rf /tmp/s1r
    SJHIP_LIB=$REPO/$lib COPIES=$c timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/s1r -o p -- python $REPO/tools/s1_device_loop.py 30 > /tmp/s1r.log 2>&1
    echo "== $lib copies=$c"; python $REPO/tools/summarize_prof.py /tmp/s1r /tmp/s1r/summary.txt > /dev/null; grep -E "kernel " /tmp/s1r/summary.txt | cut -c1-140
  done
done
