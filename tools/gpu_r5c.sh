#!/bin/bash
# round 5: HBM and SQ counters of every kernel of the whole parse (separate PMC passes), both workloads
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/gpu_pmc_parse.sh r5c/pmc_tw twitter
bash tools/gpu_pmc_parse.sh r5c/pmc_pk parking
bash tools/gpu_sq.sh r5c/sq_tw twitter
bash tools/gpu_sq.sh r5c/sq_pk parking
rm -rf gpurun_out/r5c/*/*/p_results.db gpurun_out/r5c/*/*/*.db 2>/dev/null
