#!/bin/bash
# round 5: HBM counters and VALU instructions of every kernel of the WithCopyStrings(false) parse (separate PMC passes)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
bash tools/gpu_pmc_parse.sh r5f/pmc_pk_nocopy parking nocopy
bash tools/gpu_pmc_parse.sh r5f/pmc_tw_nocopy twitter nocopy
rm -rf gpurun_out/r5f/*/*/p_results.db gpurun_out/r5f/*/*/*.db 2>/dev/null
