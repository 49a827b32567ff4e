#!/bin/bash
# stage 1 alone: kernel time on configs[1] and on the 1 GiB document (tools/s1_time.py), stage-1 GPU tests
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
for c in 426 1700 426; do COPIES=$c timeout 120 python tools/s1_time.py 2>&1 | grep -v amdgpu.ids; done
timeout 200 python -m pytest tests/test_gpu_stage1.py -m gpu -q --maxfail=5 2>&1 | tail -3
