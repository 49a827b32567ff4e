#!/bin/bash
# timing-only experiments (results are WRONG with dbg != 0): usage tools/gpu_s1_dbg.sh "variants" "dbg values"
for v in ${1:-0}; do for d in ${2:-0}; do
  echo -n "variant $v dbg $d: "
  SJHIP_S1_VARIANT=$v SJHIP_S1_DBG=$d timeout 300 python tools/s1_time.py
done; done
