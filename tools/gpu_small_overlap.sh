#!/bin/bash
# Parse() of the small fixtures host -> host under the SJHIP_S2_OVERLAP modes (0/1 one stream, 2/3 string bytes on a side stream)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for m in 1 2 3; do
  for f in twitter twitterescaped canada; do
    echo -n "overlap=$m  "; SJHIP_S2_OVERLAP=$m python tools/small_doc_trace.py $f 300 2>&1 | grep -v amdgpu.ids
  done
done
