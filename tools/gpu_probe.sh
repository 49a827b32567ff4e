#!/bin/bash
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I simdjson-go_amd/csrc tools/probes/phasea_rate.hip -o /tmp/phasea_rate && /tmp/phasea_rate > gpurun_out/phasea_rate.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate > gpurun_out/valu_rate.txt 2>&1
cat gpurun_out/phasea_rate.txt gpurun_out/valu_rate.txt
