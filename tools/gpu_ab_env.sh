#!/bin/bash
# same-box A/B of one build under two environments: tools/gpu_ab_env.sh "<VAR=a>" "<VAR=b>" [rounds]   (wall time of the whole parse)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 ${3:-3}); do
  for e in "$1" "$2"; do
    echo -n "$e  "
    env $e timeout 300 python tools/parse_time.py 2>&1 | grep -v amdgpu.ids | head -2 | tr '\n' ';'; echo
  done
done
