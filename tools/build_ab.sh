#!/bin/bash
# tools/build_ab.sh <name> [-DFLAG ...]: an A/B build of libsjhip.so with extra flags for stage1.hip and stage2.hip
# (the other translation units are compiled once into build_ab/obj).  Output: build_ab/libsjhip_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_ab/obj
CS=simdjson-go_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC"
for s in api parse_api stream_api multi_api batch_api query serialize marshal; do
  if [ ! -f build_ab/obj/$s.o ] || [ $CS/$s.hip -nt build_ab/obj/$s.o ] || [ -n "$(find $CS include -name '*.h' -newer build_ab/obj/$s.o | head -1)" ]; then
    /opt/rocm/bin/hipcc $FL -c $CS/$s.hip -o build_ab/obj/$s.o &
  fi
done
/opt/rocm/bin/hipcc $FL "$@" -c $CS/stage1.hip -o build_ab/obj/stage1_$name.o &
/opt/rocm/bin/hipcc $FL "$@" -c $CS/stage2.hip -o build_ab/obj/stage2_$name.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/libsjhip_$name.so build_ab/obj/stage1_$name.o build_ab/obj/stage2_$name.o build_ab/obj/{api,parse_api,stream_api,multi_api,batch_api,query,serialize,marshal}.o
echo built build_ab/libsjhip_$name.so
