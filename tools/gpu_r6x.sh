#!/bin/bash
# one gpurun call of the round-6 re-entry: GPU tests on the tree's library ($TESTS: default the whole suite), then same-box A/B
# against build_ab/libsjhip_head.so (the library of the last commit): plain stage 1 at three sizes, the whole parse of both
# workloads, small documents, the bench step
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest ${TESTS:-tests} -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r6x_suite.txt
NEW=simdjson-go_amd/libsjhip.so; OLD=${OLD:-build_ab/libsjhip_head.so}
{
bash tools/gpu_s1_ab.sh $OLD $NEW
bash tools/gpu_ab.sh $OLD $NEW 2
for r in 1 2; do for lib in $OLD $NEW; do
  echo -n "$lib bench step: "; SJHIP_LIB=$PWD/$lib SJHIP_BENCH_PMC=0 timeout 300 python bench.py --stage1-only --steps 30 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))"
done; done
} 2>&1 | tee gpurun_out/r6x_ab.txt
