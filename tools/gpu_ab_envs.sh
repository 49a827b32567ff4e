#!/bin/bash
# same-box A/B of one build under several environments: ROUNDS=3 tools/gpu_ab_envs.sh "<VAR=a>" "<VAR=b>" ...   (wall time of the
# whole parse, both workloads; "-" = no variable)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 ${ROUNDS:-3}); do
  for e in "$@"; do
    echo -n "$e  "
    if [ "$e" = "-" ]; then ee="SJHIP_NOTHING=1"; else ee="$e"; fi
    env $ee timeout 300 python tools/parse_time.py 2>&1 | grep -v amdgpu.ids | head -2 | awk '{printf "%s %s ms; ", $1, $(NF-3)}'; echo
  done
done
