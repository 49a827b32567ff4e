#!/bin/bash
# usage: tools/profile_s1_pmc.sh <outdir> <variant> "<counters...>" ["<counters pass 2>" ...]
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$1
export SJHIP_S1_VARIANT=$2
shift 2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o s1 -- python $REPO/tools/s1_time.py > $OUT/p$i.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "stage1_kernel" $OUT/summary.txt | sed 's/void sj::stage1_kernel<\([0-9, ]*\)>.*un /k<\1> /'
