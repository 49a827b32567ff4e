#!/bin/bash
# the whole GPU suite on the product build and on the bounds-checked build, then the soak on both: gpurun_out/final_*.txt
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/final_suite.txt
timeout 1500 bash tools/gpu_debug_bounds.sh > gpurun_out/final_debug_bounds.txt 2>&1
timeout 400 python tools/gpu_soak.py ${SOAK:-240} 2>&1 | tail -2 > gpurun_out/final_soak.txt
SJHIP_LIB=$PWD/simdjson-go_amd/libsjhip_dbg.so timeout 300 python tools/gpu_soak.py 120 2>&1 | tail -2 >> gpurun_out/final_soak.txt
cat gpurun_out/final_suite.txt gpurun_out/final_debug_bounds.txt gpurun_out/final_soak.txt | grep -v amdgpu.ids
