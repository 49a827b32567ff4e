#!/bin/bash
# sweep-order experiment (SJ_EXP bits 20: k_str_emit from the last unit to the first, 21: the emit pass from the last tile to the first)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export SJHIP_LIB=$PWD/build_ab/libsjhip_exp.so
OUT=gpurun_out/r6s_sweep.txt
: > $OUT
ROUNDS=4 WORKLOADS=twitter,parking,twitter1g timeout 600 python tools/exp_time.py 0 0x100000 0x200000 0x300000 0 2>&1 | grep -v amdgpu.ids | tee -a $OUT
MODE=nocopy ROUNDS=3 WORKLOADS=twitter,parking timeout 300 python tools/exp_time.py 0 0x100000 0x200000 0x300000 2>&1 | grep -v amdgpu.ids | tee -a $OUT
