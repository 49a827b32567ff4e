#!/bin/bash
# The GPU test suite on the bounds-checked debug build (csrc/sj_bounds.h): tools/gpu_debug_bounds.sh [test files; default: the whole GPU suite]
# Build libsjhip_dbg.so first (python -c "import __graft_entry__ as g; g.build_lib(debug_bounds=True)"): it travels with the push.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
export SJHIP_LIB=$REPO/simdjson-go_amd/libsjhip_dbg.so
python - <<'PY'
import sys
sys.path.insert(0, "simdjson-go_amd")
import sjhip
print("debug build self-test (2 = both deliberate violations recorded):", sjhip.lib().sjhip_debug_bounds_selftest())
PY
if [ $# -gt 0 ]; then python -m pytest "$@" -m gpu -q 2>&1 | tail -25; else python -m pytest tests -m gpu -q 2>&1 | tail -25; fi
