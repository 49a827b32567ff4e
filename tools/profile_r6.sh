#!/bin/bash
# Round-6 profile (GPU box, repo root): tools/profile_r6.sh <name>  -> gpurun_out/<name>/
#  (1) bench.py plain (the bench line), (2) kernel trace + stats of the whole parse on the two BASELINE workloads (both copy modes),
#  (3) HBM counters of every parse kernel (FETCH_SIZE / WRITE_SIZE in separate PMC passes, MI355X_MICROARCH.md),
#  (5) MarshalJSON / serializer kernels on configs[4]'s tape,
#  (4) stage 1 alone: kernel trace + FETCH/WRITE passes on the 64 MiB document (x107), configs[1] and the 1 GiB document (x1700).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-prof_r6}
mkdir -p $OUT
cd $REPO
timeout 900 python bench.py > $OUT/bench_plain.log 2>&1
grep -h "^{" $OUT/bench_plain.log | tail -1 > $OUT/bench.json
cd /tmp && export TMPDIR=/tmp
for w in twitter parking; do
  timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/trace_$w -o p -- python $REPO/tools/parse_loop.py $w 6 > $OUT/trace_$w.log 2>&1
  timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$w -o p -- python $REPO/tools/parse_loop.py $w 3 > $OUT/fetch_$w.log 2>&1
  timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write_$w -o p -- python $REPO/tools/parse_loop.py $w 3 > $OUT/write_$w.log 2>&1
done
# (2b) the same parses with WithCopyStrings(false)
for w in twitter parking; do
  timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/trace_nocopy_$w -o p -- python $REPO/tools/parse_loop.py $w 6 nocopy > $OUT/trace_nocopy_$w.log 2>&1
done
# (2c) small documents: kernel timelines (device-resident and host -> host)
for f in twitter twitterescaped canada; do
  for tool in small_dev_loop small_doc_trace; do
    rm -rf $OUT/t_small
    timeout 120 rocprofv3 --kernel-trace -d $OUT/t_small -o p -- python $REPO/tools/$tool.py $f 10 > $OUT/small.log 2>&1
    python $REPO/tools/timeline.py $OUT/t_small/p_results.db 1 > $OUT/timeline_${f}_$tool.txt 2>&1
  done
done
rm -rf $OUT/t_small
# (5) MarshalJSON / serializer of configs[4]'s tape (key flags from the parser: the single-pass form)
timeout 90 rocprofv3 --kernel-trace --stats -d $OUT/trace_marshal -o p -- python $REPO/tools/marshal_loop.py parking 5 kf > $OUT/trace_marshal.log 2>&1
for c in 107 426 1700; do
  export COPIES=$c
  timeout 60 rocprofv3 --kernel-trace --stats -d $OUT/s1trace_$c -o s1 -- python $REPO/tools/s1_time.py > $OUT/s1trace_$c.log 2>&1
  timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/s1fetch_$c -o s1 -- python $REPO/tools/s1_time.py > $OUT/s1fetch_$c.log 2>&1
  timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/s1write_$c -o s1 -- python $REPO/tools/s1_time.py > $OUT/s1write_$c.log 2>&1
done
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -E "^==|pmc|kernel " $OUT/summary.txt | sed "s/(sj::S2Dev[^)]*)//" | grep -v fillBuffer | cut -c1-200 | head -150
cat $OUT/bench.json | cut -c1-3000
rm -rf $OUT/*/*.db $OUT/*/*/*.db 2>/dev/null
