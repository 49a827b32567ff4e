#!/bin/bash
# builds and runs the probes named on the command line (tools/probes/<name>.hip) on the GPU box
mkdir -p gpurun_out
cd "$(dirname "$0")/.."
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I simdjson-go_amd/csrc tools/probes/$n.hip -o /tmp/$n && timeout 120 /tmp/$n > gpurun_out/$n.txt 2>&1
  cat gpurun_out/$n.txt
done
