#!/bin/bash
# the queued stage-1 API: its test, then the bench's stage-1 step (queued vs synchronised) three times
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_stage1.py -m gpu -x -q 2>&1 | tail -4
for r in 1 2 3; do
  SJHIP_BENCH_PMC=0 timeout 300 python bench.py --stage1-only --steps 20 --warmup 3 2>gpurun_out/r6q_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('queued', d['value'], d['ms_per_step'], 'synchronised', d['value_synchronised'], d['ms_per_step_synchronised'], 'kernel', d['roofline'].get('kernel_ms'))" || tail -5 gpurun_out/r6q_err.txt
done
SJHIP_BENCH_PMC=0 timeout 300 python bench.py --stage1-only --steps 200 --warmup 3 2>gpurun_out/r6q_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('200 steps: queued', d['value'], d['ms_per_step'], 'synchronised', d['value_synchronised'], d['ms_per_step_synchronised'], 'kernel', d['roofline'].get('kernel_ms'))" || tail -5 gpurun_out/r6q_err.txt
} 2>&1 | tee gpurun_out/r6q.txt
