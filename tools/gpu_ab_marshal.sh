#!/bin/bash
# A/B of MarshalJSON wall time between the libraries in build_ab/ (same box, interleaved twice).  The variants are built by
# hand into build_ab/libsjhip_ms_<name>.so (the objects of __graft_entry__.HIP_SOURCES with marshal.hip compiled per
# variant; build_ab/ is git-ignored and travels with gpurun); SJHIP_LIB selects the library the Python mirror loads.
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
for lib in $(ls build_ab/libsjhip_ms_*.so); do
for w in ${MS_WORKLOADS:-parking twitter}; do
echo -n "$lib "; SJHIP_LIB=$PWD/$lib timeout 200 python tools/marshal_loop.py $w 5 kf 2>&1 | grep marshal_json
done; done; done
