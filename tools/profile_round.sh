#!/bin/bash
# Round profile: (1) rocprofv3 kernel trace + stats of the default bench command, (2) PMC passes on the stage-1 kernel
# (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes), (3) SQ instruction counters.
# Usage (GPU box, repo root): tools/profile_round.sh <name>      -> gpurun_out/<name>/summary.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-prof_round}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline"
S1="python $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --stage1-only"
# the bench line itself from an unprofiled run of the same command (the tracer adds launch overhead to the hipEvent timings)
$BENCH > $OUT/bench_plain.log 2>&1
grep -h "^{" $OUT/bench_plain.log | tail -1 > $OUT/bench.json
timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $BENCH > $OUT/trace.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o s1 -- $S1 > $OUT/pmc_fetch.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o s1 -- $S1 > $OUT/pmc_write.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o s1 -- $S1 > $OUT/pmc_sq.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o s1 -- $S1 > $OUT/pmc_sq2.log 2>&1
cd $REPO && python tools/summarize_prof.py $OUT $OUT/summary.txt > /dev/null
grep -h "^{" $OUT/trace.log | tail -1 > $OUT/bench_traced.json
python tools/make_pmc_json.py $OUT/summary.txt $OUT/stage1_pmc.json "profiles/${1:-prof_round}_bench_rocprofv3_summary.txt (tools/profile_round.sh)" > /dev/null
