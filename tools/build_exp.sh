#!/bin/bash
# tools/build_exp.sh: build_ab/libsjhip_exp.so = the library with -DSJ_EXP (parts of the whole-parse kernels can be left
# out through the SJHIP_EXP bit mask: A/B timing of what each part costs; results are wrong by construction)
set -e
cd "$(dirname "$0")/.."
mkdir -p build_ab
CS=simdjson-go_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -DSJ_EXP -o build_ab/libsjhip_exp.so \
  $CS/stage1.hip $CS/stage2.hip $CS/api.hip $CS/parse_api.hip $CS/stream_api.hip $CS/multi_api.hip $CS/batch_api.hip $CS/query.hip $CS/serialize.hip $CS/marshal.hip
echo built build_ab/libsjhip_exp.so
