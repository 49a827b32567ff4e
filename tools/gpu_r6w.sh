#!/bin/bash
# tests ($TESTS) on the tree's library, then same-box A/B for the libraries in $LIBS: plain stage 1 at three sizes (kernel time incl.
# its epilogue), the whole parse in both copy modes, small documents, the bench's stage-1 step
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
if [ -n "${TESTS:-}" ]; then timeout 1500 python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r6w_suite.txt; fi
{
bash tools/gpu_s1_ab.sh $LIBS
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in $LIBS; do
    echo -n "$lib  "
    SJHIP_LIB=$PWD/$lib timeout 300 python tools/nocopy_time.py 2>&1 | grep -v amdgpu.ids | awk '{printf "%s/%s %s; ", $1, $2, $3}'
    echo
    SJHIP_LIB=$PWD/$lib timeout 300 python tools/small_time.py 2>&1 | grep -v amdgpu.ids | tail -1
  done
done
for r in 1 2; do for lib in $LIBS; do
  echo -n "$lib bench step: "; SJHIP_LIB=$PWD/$lib SJHIP_BENCH_PMC=0 timeout 300 python bench.py --stage1-only --steps 30 --warmup 5 2>gpurun_out/r6w_bench_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms'))" || tail -5 gpurun_out/r6w_bench_err.txt
done; done
} 2>&1 | tee gpurun_out/r6w_ab.txt
