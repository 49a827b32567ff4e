#!/bin/bash
# host -> host breakdown of Parse() (tools/h2h_bench.cpp) on the fixtures BASELINE.json names
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-h2h}
mkdir -p $OUT
cd $REPO
g++ -O2 -std=c++17 -w tools/h2h_bench.cpp -I include -I /opt/rocm/include -D__HIP_PLATFORM_AMD__ -L simdjson-go_amd -lsjhip \
    -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$REPO/simdjson-go_amd -o /tmp/h2h_bench || exit 1
python - <<'PY'
import sys
sys.path.insert(0, "tests")
import fixtures
for n in ("twitter", "canada", "twitterescaped", "parking-citations"):
    open(f"/tmp/{n}.json", "wb").write(fixtures.load(n))
PY
for n in twitter canada twitterescaped; do timeout 120 /tmp/h2h_bench /tmp/$n.json 2>&1 | grep -v amdgpu.ids | tee -a $OUT/h2h.txt; done
timeout 120 /tmp/h2h_bench /tmp/parking-citations.json nd 2>&1 | grep -v amdgpu.ids | tee -a $OUT/h2h.txt
