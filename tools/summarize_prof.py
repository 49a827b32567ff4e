#!/usr/bin/env python3
"""Condense rocprofv3 (ROCm 7.2, rocpd sqlite output) result directories into one text summary:
per-kernel call count / average duration (view top_kernels) and per-kernel PMC counters
(view counters_collection: summed over hardware instances per dispatch, averaged over dispatches)."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def main(root, out):
    lines = []
    for db in sorted(glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)):
        con = sqlite3.connect(db)
        cur = con.cursor()
        lines.append(f"== {os.path.relpath(db, root)}")
        try:
            for name, calls, total, avg, pct in cur.execute(
                    "select name,total_calls,total_duration,average,percentage from top_kernels"):
                lines.append(f"  kernel {name[:70]:70s} calls={calls} avg_us={avg:.3f} total_us={total:.1f} pct={pct:.1f}")
        except sqlite3.Error as e:
            lines.append(f"  (no top_kernels: {e})")
        try:
            per = defaultdict(lambda: defaultdict(float))
            for kname, cname, disp, val in cur.execute(
                    "select kernel_name,counter_name,dispatch_id,value from counters_collection"):
                per[(kname[:70], cname)][disp] += val
            for (kname, cname), d in sorted(per.items()):
                vals = list(d.values())
                lines.append(f"  pmc {kname:70s} {cname:24s} dispatches={len(vals)} mean={sum(vals)/len(vals):.1f} "
                             f"min={min(vals):.1f} max={max(vals):.1f}")
        except sqlite3.Error as e:
            lines.append(f"  (no counters: {e})")
    text = "\n".join(lines) + "\n"
    open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
