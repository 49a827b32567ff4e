#!/bin/bash
# same-box per-kernel A/B of two builds of the library: tools/gpu_ab_kernels.sh <a.so> <b.so> [twitter|parking] [copy|nocopy]
# (rocprofv3 kernel trace of tools/parse_loop.py, median duration per kernel in us)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/abk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=${3:-twitter}; M=${4:-copy}
for r in 1 2; do
for lib in $1 $2; do
  rm -rf $OUT/t
  SJHIP_LIB=$REPO/$lib timeout 120 rocprofv3 --kernel-trace -d $OUT/t -o p -- python $REPO/tools/parse_loop.py $W 6 $M > $OUT/log 2>&1
  echo "== $lib ($W $M, round $r)"
  python - $OUT/t/p_results.db <<'PY'
import sqlite3, sys
from collections import defaultdict
con = sqlite3.connect(sys.argv[1])
d = defaultdict(list)
for n, dur in con.execute("select name, (end - start) from kernels order by start"):
    d[n.split("(")[0].replace("void ", "").replace("sj::", "")[:40]].append(dur / 1000.0)
tot = 0
for n, v in d.items():
    v = sorted(v[len(v) // 3:])  # (skip the warm-up third)
    m = v[len(v) // 2]
    tot += m
    print(f"   {n:40s} {m:7.1f}")
print(f"   {'sum':40s} {tot:7.1f}")
PY
done
done
