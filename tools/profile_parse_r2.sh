#!/bin/bash
# Whole-parse profile of the two BASELINE workloads: kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in separate PMC
# passes (MI355X_MICROARCH.md).  Usage (GPU box, repo root): tools/profile_parse_r2.sh <name> -> gpurun_out/<name>/
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-prof_parse_r2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in twitter parking; do
  timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o p -- python $REPO/tools/parse_loop.py $w 6 > $OUT/trace_$w.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$w -o p -- python $REPO/tools/parse_loop.py $w 3 > $OUT/fetch_$w.log 2>&1
  timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$w -o p -- python $REPO/tools/parse_loop.py $w 3 > $OUT/write_$w.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
python tools/make_parse_json.py $OUT/summary.txt $OUT/r02_parse_kernels.json $OUT/stage2_pmc.json
