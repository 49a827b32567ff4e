#!/bin/bash
# A/B of the parts of the whole-parse kernels (build_ab/libsjhip_exp.so, tools/build_exp.sh): kernel times with one
# part left out at a time.  -> gpurun_out/$1/exp.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-exp}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export SJHIP_LIB=$REPO/build_ab/libsjhip_exp.so
: > $OUT/exp.txt
for w in twitter parking; do
for bits in 0 256 512 1024 2048 4096 8192 65536 131072 262144 458752 32 64 0; do
  rm -rf $OUT/t
  SJHIP_EXP=$bits timeout 120 rocprofv3 --kernel-trace -d $OUT/t -o p -- python $REPO/tools/parse_loop_noexc.py $w 3 > $OUT/log 2>&1
  echo "== $w exp=$bits" >> $OUT/exp.txt
  python $REPO/tools/kernel_times.py $OUT/t/p_results.db 2>/dev/null | grep -E "stage1_kernel|k_str_emit|k_s2_emit|k_str_masks|k_numbers|k_br_match" | awk '{n=split($0,a,"us:"); print $1, a[2]}' >> $OUT/exp.txt
done
done
cat $OUT/exp.txt
